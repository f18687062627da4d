"""The latency floor of one pitch step of the walk kernel -- the denominator-side model of bench.py's `roofline.latency`.

A stream's walk is a chain of dependent pitch steps (the position of step k+1 is the period step k found); with one stream per
CU the chain IS the run time, so the honest roofline for BASELINE configs[3] is "how many shader cycles must one step take",
priced with the measured costs of single dependent operations on gfx950 (tools/ubench/issue_costs.hip; the table below is
its output for 8 waves per CU of which 4 execute -- the walk kernel's 4 search + 4 output waves -- profiles/r03/r03z_issue_costs.txt,
re-taken in round 4: profiles/r04/).  The floor counts ONLY what is dependent by the algorithm's data flow:

  coarse search   addresses of the lane's operands from the step's window offset      (dependent VALU chain)
                  the operand loads, all in flight                                     (one LDS burst)
                  masked SADs of the last-arriving operands, lag sums met by ds_add    (VALU + LDS atomic round trip)
                  workgroup barrier, sums read back                                    (barrier + LDS round trip)
                  arg-min of diff / lag: key, 6-stage DPP min, ballot, first bit        (DPP reduction)
  refine search   lag window from the winner, per-lane rectangle addresses              (SALU + dependent VALU chain)
                  operand loads in flight (up to 15 reads per lane)                     (one LDS burst, bandwidth-bound)
                  SADs of the last operands, ds_add, barrier, read back, arg-min        (as above)
                  previous-period rule, n = readlane(candidate division), position      (v_readlane -> SALU hops)
  between steps   event bookkeeping (can this event run another step, else the next)   (SALU chain + one ballot)

Everything else a step executes (publishing the command, the candidate divisions, zeroing the other sum buffer, the ragged
tasks, cross-fades, refills) is independent work that an ideal schedule hides in the shadows of the round trips above.

    python tools/latency_model.py            # prints the table, writes profiles/latency_model.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# shader cycles, gfx950, 8 waves per CU of which 4 execute (tools/ubench/issue_costs.hip)
COST = {
    "salu": 4.1,            # dependent s_add_i32
    "valu": 5.5,            # dependent v_add_u32
    "lds_read": 67.6,       # ds_read_b32 -> wait -> use
    "lds_burst8": 127.1,    # 8 x ds_read2_b32 in flight + wait
    "lds_atomic": 56.1,     # ds_add_u32 + wait
    "barrier": 13.0,        # s_barrier, all waves already there
    "dpp_min": 97.8,        # 6-stage DPP min + readlane + v_add
    "readlane_hop": 28.1,   # v_readlane -> s_add -> v_add (one VALU -> SALU -> VALU hop)
    "fma64": 5.4,           # dependent v_fma_f64
    "sad": 6.3,             # dependent v_sad_u16
}

# (phase, item, how many of which cost)
ITEMS = [
    ("coarse", "operand addresses: o/skip, plane, two pair_addr (dependent VALU)", 8, "valu"),
    ("coarse", "operand loads: 2 groups x 2 operands x 2 ds_read2_b32 in flight", 1, "lds_burst8"),
    ("coarse", "SADs of the last group: 2 v_and + v_sad per slot, 4 slots", 4, "sad"),
    ("coarse", "lag sums meet: ds_add_u32", 1, "lds_atomic"),
    ("coarse", "workgroup barrier", 1, "barrier"),
    ("coarse", "sums read back", 1, "lds_read"),
    ("coarse", "arg-min: cvt + fma key", 2, "fma64"),
    ("coarse", "arg-min: DPP min, ballot, first set bit", 1, "dpp_min"),
    ("refine", "lag window: period, lo / hi clamps, nl, c0, G, NGL (SALU)", 10, "salu"),
    ("refine", "per-lane rectangle addresses: pT, ea, two pair_addr, left-over offsets (dependent VALU)", 10, "valu"),
    ("refine", "operand loads: up to 15 reads per lane in flight = 2 bursts of the LDS pipe at 4 waves", 2, "lds_burst8"),
    ("refine", "SADs of the last flight", 4, "sad"),
    ("refine", "lag sums meet: ds_add_u32", 1, "lds_atomic"),
    ("refine", "workgroup barrier", 1, "barrier"),
    ("refine", "sums read back", 1, "lds_read"),
    ("refine", "arg-min: table scale (LDS, issued early) + fma key", 2, "fma64"),
    ("refine", "arg-min: DPP min, ballot, first set bit", 1, "dpp_min"),
    ("refine", "previous-period rule test (scalar compares)", 4, "salu"),
    ("refine", "n = readlane(candidate divisions): one VALU -> SALU hop", 1, "readlane_hop"),
    ("between", "position, output count, overflow test, loop condition (SALU)", 8, "salu"),
    ("between", "next runnable event: ballot over 64 events + first bit + speed readlane", 1, "readlane_hop"),
]


def main():
    phases = {}
    rows = []
    for ph, what, n, key in ITEMS:
        c = n * COST[key]
        phases[ph] = phases.get(ph, 0.0) + c
        rows.append({"phase": ph, "item": what, "count": n, "unit": key, "cycles": round(c, 1)})
    floor = sum(phases.values())
    model = {"_note": "tools/latency_model.py: dependent-latency floor of one pitch step (two dependent searches) on gfx950, from "
                      "tools/ubench/issue_costs.hip constants (8 waves per CU, 4 executing); DESIGN.md 6",
             "costs_cycles": COST, "items": rows, "phases_cycles": {k: round(v, 1) for k, v in phases.items()},
             "floor_cycles_per_step": round(floor, 1)}
    for r in rows:
        print("%-8s %-88s %2d x %-12s = %6.1f" % (r["phase"], r["item"], r["count"], r["unit"], r["cycles"]))
    print("phases:", model["phases_cycles"], " floor: %.0f cycles per step" % floor)
    if "--no-write" not in sys.argv:
        with open(os.path.join(ROOT, "profiles", "latency_model.json"), "w") as fh:
            json.dump(model, fh, indent=1)
        print("wrote profiles/latency_model.json")


if __name__ == "__main__":
    main()
