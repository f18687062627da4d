"""The engine's launch-mode decision on the CPU (round 5).  spx_choose_mode (speedy_amd/csrc/spx_mode.h) is a pure function of
the batch's shape, the kernels' resources, the process settings and the state of the plan's ring / trial; run_impl only executes
what it decides.  Here it is compiled by plain g++ (speedy_amd/csrc/spx_mode_table.cpp -> speedy_amd/lib/libspx_mode_table.so) and
fed the resource numbers of the shipped kernels (profiles/kernel_resources.json, written on the GPU box by
tools/kernel_resources.py and checked against the library by tests/test_gpu_parity.py): every (rate, channels, streams, entry
point, ring state) the GPU tests and the bench rely on has its row -- and so has "the tension kernel takes 49 registers", the
change that silently switched a mode off in round 3."""
import ctypes as C
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "speedy_amd", "lib", "libspx_mode_table.so")


@pytest.fixture(scope="module")
def table():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "modetable"])
    L = C.CDLL(LIB)
    L.spx_mode_table_query_fields.restype = C.c_char_p
    L.spx_mode_table_answer_fields.restype = C.c_char_p
    qf = L.spx_mode_table_query_fields().decode().split()
    af = L.spx_mode_table_answer_fields().decode().split()
    res = json.load(open(os.path.join(ROOT, "profiles", "kernel_resources.json")))
    assert res["fields"] == qf[qf.index("cu_count"): qf.index("an_vgprs_small") + 1], "profiles/kernel_resources.json and spx_mode_table.cpp disagree about the fields"

    def decide(shape, **kw):
        """shape: 'rate,channels,streams,speedup_only' (a key of kernel_resources.json); kw: any query field."""
        q = dict.fromkeys(qf, 0)
        q.update(dict(zip(res["fields"], res["shapes"][shape])))
        rate, ch, n, sp = (int(v) for v in shape.split(","))
        q.update(n=n, max_channels=ch, do_a=1, do_w=1, has_frames=1, force_total_streams=n, concurrent_enabled=1, chunks=1,
                 trial_force=-1, trial_state_key=-1, trial_choice=-1, trial_key=77, device_ours=1)
        q.update(kw)
        qa = (C.c_longlong * len(qf))(*[int(q[f]) for f in qf])
        aa = (C.c_longlong * len(af))()
        assert L.spx_mode_table_eval(qa, len(qf), aa, len(af)) == 0
        return dict(zip(af, [int(v) for v in aa]))

    decide.lib = L
    decide.res = res
    return decide


def kind(a):
    return "concurrent" if a["concurrent"] else ("ahead" if a["ahead"] else "sequence")


def test_headline_shape_16k_mono_256(table):
    a = table("16000,1,256,1")                                   # spx_batch_run: three kernels side by side, 16-frame tile, full walk form
    assert (kind(a), a["tile_frames"], a["launch_lean"], a["doubtful"], a["nch"], a["exclusive_cu"]) == ("concurrent", 16, 0, 0, 1, 0)
    a = table("16000,1,256,1", ahead_req=1)                      # spx_batch_run_ahead: pipelined, walk on the caller's stream
    assert (kind(a), a["seq_ahead"], a["walk2"], a["launch_lean"]) == ("ahead", 0, 0, 0)
    a = table("16000,1,256,1", ahead_req=1, overlap_req=1)       # spx_batch_run_overlapped / the pipeline, three or more buffer sets: LEAN
    assert (kind(a), a["walk2"], a["launch_lean"], a["lean_walk"]) == ("ahead", 1, 1, 1)
    a = table("16000,1,256,1", ahead_req=1, overlap_req=1, two_workspaces=1)   # two buffer sets taking turns: the full form stays
    assert (kind(a), a["walk2"], a["launch_lean"]) == ("ahead", 1, 0)
    for n in ("16000,1,97,1", "16000,1,64,1"):
        assert kind(table(n)) == "concurrent" and kind(table(n, ahead_req=1)) == "ahead"


def test_the_lean_form_is_decided_after_the_pipelined_mode(table):
    """Round-4 advisor: an overlapped call that falls out of the pipelined mode must not run the lean form where the full one fits."""
    a = table("16000,1,256,1", ahead_req=1, overlap_req=1, chunks_set=1, chunks=2)      # the caller chose time chunks: no pipelining
    assert kind(a) != "ahead" and a["launch_lean"] == 0 and a["lean_walk"] == 0
    a = table("16000,1,256,1", ahead_req=1, overlap_req=1, concurrent_enabled=0)         # spx_set_concurrent(0)
    assert (kind(a), a["launch_lean"], a["exclusive_cu"]) == ("sequence", 0, 1)
    a = table("16000,1,256,1", ahead_req=1, overlap_req=1, device_ours=0)                # another process holds the device's lock
    assert (kind(a), a["launch_lean"], a["asked_device"]) == ("sequence", 0, 1)


def test_22k_mono_needs_the_lean_form_and_every_register(table):
    a = table("22050,1,256,1")
    assert (kind(a), a["launch_lean"], a["tile_frames"], a["doubtful"]) == ("concurrent", 1, 16, 0)      # 128 + 48 + 2 x 168 = 512, exact
    assert kind(table("22050,1,256,1", ahead_req=1)) == "ahead"
    # ... ONE more register in the tension kernel (49 -> an allocation of 56) and the shape is a doubtful one: full walk form,
    # 8-frame tile, a timed trial instead of the mode.  The same for the lean walk kernel or the analysis kernel growing.
    r = dict(zip(table.res["fields"], table.res["shapes"]["22050,1,256,1"]))
    for grow in (dict(tension_vgprs=r["tension_vgprs"] + 8), dict(lean_vgprs=r["lean_vgprs"] + 8),
                 dict(an_vgprs_default=r["an_vgprs_default"] + 8, an_vgprs_small=r["an_vgprs_small"] + 8)):
        a = table("22050,1,256,1", **grow)
        assert a["doubtful"] == 1 or not a["concurrent"], grow
        assert a["launch_lean"] == 0 or "an_vgprs_default" in grow, grow
    # the headline shape has 16 registers to spare: the same growth does not move it
    assert kind(table("16000,1,256,1", tension_vgprs=56)) == "concurrent"


def test_multi_channel_shapes(table):
    a = table("16000,2,256,1")      # 2 x 96 + 48 + 2 x 128 = 496: the concurrent mode, no trial
    assert (kind(a), a["doubtful"], a["launch_lean"]) == ("concurrent", 0, 0)
    assert kind(table("16000,2,128,1", ahead_req=1, overlap_req=1)) == "ahead"
    # 22.05 kHz stereo, 256 streams: the consumers' workgroups could close every CU to the analysis kernel (rule 1): in sequence
    assert kind(table("22050,2,256,1")) == "sequence"
    # ... 128 streams: rule 1 holds, but only ONE analysis wave fits beside the walk waves -> a trial decides between concurrent and sequence
    seq = []
    state = dict(trial_state_key=-1, trial_calls=0, trial_choice=-1)
    for call in range(5):
        a = table("22050,2,128,1", **state, trial_times_ready=1 if call >= 3 else 0, us_seq=2280, us_con=3100)
        seq.append((kind(a), a["trial_slot"]))
        state = dict(trial_state_key=a["next_key"], trial_calls=a["next_calls"], trial_choice=a["next_choice"])
    assert seq == [("concurrent", -1), ("concurrent", 1), ("sequence", 0), ("sequence", -1), ("sequence", -1)]
    assert state["trial_choice"] == 0
    # ... and once it runs in sequence it can still be pipelined with its predecessor (one analysis workgroup beside a walk
    # workgroup that keeps its CU): seq_ahead, walk kernel on the caller's stream, exclusive CUs
    a = table("22050,2,256,1", ahead_req=1, overlap_req=1, trial_state_key=77, trial_calls=5, trial_choice=0)
    assert (kind(a), a["seq_ahead"], a["walk2"], a["exclusive_cu"], a["tile_frames"]) == ("ahead", 1, 0, 1, 16)
    # while the trial is still running the call is not pipelined
    a = table("22050,2,128,1", ahead_req=1, trial_state_key=77, trial_calls=1, trial_choice=-1)
    assert a["ahead"] == 0 and a["trial_slot"] == 1
    a = table("22050,2,128,1", ahead_req=1, trial_state_key=77, trial_calls=5, trial_choice=0)
    assert (kind(a), a["seq_ahead"]) == ("ahead", 1)


def test_shapes_outside_every_mode(table):
    for shape in ("44100,1,256,1", "48000,2,256,1", "48000,1,64,1"):     # the analysis tiles do not fit beside the walk workgroups
        for kw in (dict(), dict(ahead_req=1), dict(ahead_req=1, overlap_req=1)):
            a = table(shape, **kw)
            assert (kind(a), a["exclusive_cu"], a["launch_lean"]) == ("sequence", 1, 0), (shape, kw)
    # slow-down jobs (round 5: the speed-up kernel's plan-driven instantiation with the insertPitchPeriod event, 128 registers):
    # two of its waves per SIMD leave no room for two analysis waves, its lean form does -- the concurrent mode with the lean form
    a = table("16000,1,256,0")
    assert (kind(a), a["launch_lean"], a["doubtful"]) == ("concurrent", 1, 0)
    # more streams than CUs: kernels in sequence; the throughput regime (more than two per CU) in two time chunks
    a = table("16000,1,512,1")
    assert (kind(a), a["nch"], a["exclusive_cu"]) == ("sequence", 1, 0)
    a = table("16000,1,2048,1")
    assert (kind(a), a["nch"]) == ("sequence", 2)
    assert table("16000,1,2048,1", chunks_set=1, chunks=1)["nch"] == 1
    assert kind(table("16000,1,600,1", ahead_req=1, overlap_req=1)) == "sequence"


def test_separate_halves_guard_and_forced_groups(table):
    assert kind(table("16000,1,256,1", do_w=0)) == "sequence" and kind(table("16000,1,256,1", do_a=0)) == "sequence"
    # another concurrent-mode call in flight on the device: this one runs in sequence
    a = table("16000,1,256,1", guard_busy=1)
    assert (kind(a), a["want_concurrent"], a["tile_frames"]) == ("sequence", 0, 16)
    # a batch without a single analysis frame has nothing to hand over -- and nothing to run ahead of anything
    assert table("16000,1,256,1", has_frames=0)["concurrent"] == 0
    a = table("16000,1,256,1", has_frames=0, ahead_req=1, overlap_req=1)
    assert (kind(a), a["walk2"], a["launch_lean"]) == ("sequence", 0, 0)
    # a group of a mixed-rate call takes what was decided for all groups together
    a = table("16000,2,128,1", forced=1, force_concurrent=0, force_ahead=1, force_total_streams=256, ahead_req=0)
    assert (kind(a), a["ahead_forced"], a["walk2"], a["exclusive_cu"], a["asked_device"]) == ("ahead", 1, 0, 1, 0)
    a = table("16000,2,128,1", forced=1, force_concurrent=1, force_total_streams=256)
    assert kind(a) == "concurrent"
    a = table("16000,2,128,1", forced=1, force_total_streams=700)
    assert (kind(a), a["exclusive_cu"]) == ("sequence", 0)


def test_mixed_rate_call(table):
    """spx_choose_mixed_mode on BASELINE configs[4]'s shard: 128 streams at 16 kHz, 128 at 22.05 kHz, half of each stereo."""
    L = table.lib
    r16 = dict(zip(table.res["fields"], table.res["shapes"]["16000,2,128,1"]))
    r22 = dict(zip(table.res["fields"], table.res["shapes"]["22050,2,128,1"]))
    groups = []
    for r in (r16, r22):
        groups += [128, r["walk_lds"], r["walk_waves"], r["walk_vgprs"], 1, r["an_lds_default"], r["an_vgprs_default"]]
    g = (C.c_longlong * len(groups))(*groups)
    out = (C.c_longlong * 4)()

    def mixed(n_total=256, enabled=1, serial=0, ahead=0, busy=0, ours=1, env_mixed=-1):
        assert L.spx_mode_table_eval_mixed(g, 2, n_total, r16["cu_count"], r16["lds_per_cu"], r16["tension_lds"], r16["tension_vgprs"],
                                           enabled, serial, env_mixed, 0, ahead, busy, ours, out) == 0
        return dict(concurrent=out[0], ahead=out[1], chain=out[2], asked=out[3])
    assert mixed() == dict(concurrent=0, ahead=0, chain=1, asked=0)          # 2 x 128 + 48 + 2 x 168 > 512: kernels in sequence, analyses chained
    assert mixed(ahead=1) == dict(concurrent=0, ahead=1, chain=1, asked=1)   # spx_batch_run_mixed_ahead
    assert mixed(ahead=1, ours=0)["ahead"] == 0 and mixed(ahead=1, enabled=0)["ahead"] == 0
    assert mixed(n_total=2048, ahead=1)["ahead"] == 0
    # a mix that WOULD run concurrently (here: said so by the tuning switch) and finds another concurrent call in flight runs in
    # sequence -- and, asked to, still pipelined with its predecessor (round 5 computed `ahead` before the busy guard: ADVICE r5)
    assert mixed(env_mixed=1)["concurrent"] == 1
    assert mixed(env_mixed=1, busy=1) == dict(concurrent=0, ahead=0, chain=1, asked=1)
    assert mixed(env_mixed=1, busy=1, ahead=1) == dict(concurrent=0, ahead=1, chain=1, asked=1)
    # round 6: the walk kernels of consecutive mixed calls overlap only for a pipelined, detached call without taps
    # (spx_mixed_walk2: concurrent, ahead, detached, taps, the walk1 switch)
    w2 = L.spx_mode_table_mixed_walk2
    assert w2(0, 1, 1, 0, 0) == 1
    assert w2(0, 1, 0, 0, 0) == 0 and w2(0, 0, 1, 0, 0) == 0 and w2(1, 0, 1, 0, 0) == 0
    assert w2(0, 1, 1, 1, 0) == 0 and w2(0, 1, 1, 0, 1) == 0
