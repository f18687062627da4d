"""Batch engine: N independent streams, device-resident, one spx_batch_run call (include/speedy_hip.h).

Host-side mirror of what the reference's caller loop does per stream (speedy_wave.cc:154-242): create,
setSpeed, enableNonlinear, setFeedback, write everything, flush, drain."""
import ctypes as C

import numpy as np
import torch

from ._lib import StreamJob, Taps, lib


class Plan:
    def __init__(self, sample_rate, match_matlab=False):
        self.L = lib()
        if not torch.cuda.is_available():
            raise RuntimeError("speedy_amd needs a HIP device; there is no CPU path")
        self.h = self.L.spx_plan_create(int(sample_rate), int(bool(match_matlab)))
        if not self.h:
            raise RuntimeError("spx_plan_create: " + self.L.spx_last_error().decode())
        self.sample_rate = sample_rate
        self.B = self.L.spx_plan_frame_step(self.h)
        self.W = self.L.spx_plan_window_size(self.h)
        self.N = self.L.spx_plan_fft_size(self.h)
        self.F = self.L.spx_plan_future(self.h)
        self.max_required = self.L.spx_plan_max_required(self.h)

    def frames(self, n_in):
        return self.L.spx_plan_frames(self.h, int(n_in))

    def out_capacity(self, n_in, speed, nonlinear=1.0):
        return self.L.spx_plan_out_capacity_for(self.h, int(n_in), float(speed), float(nonlinear))

    def close(self):
        if self.h:
            self.L.spx_plan_destroy(self.h)
            self.h = None


class Batch:
    """A prepared batch: inputs packed into one HBM buffer, outputs and workspace allocated once."""

    def __init__(self, plan, lengths, channels, speed, nonlinear=1.0, feedback=0.0, device="cuda", taps=False,
                 spectrogram_taps=False):
        self.plan = plan
        n = len(lengths)
        self.n = n
        ch = np.broadcast_to(np.asarray(channels, np.int32), (n,)).copy()
        sp = np.broadcast_to(np.asarray(speed, np.float32), (n,)).copy()
        nlv = np.broadcast_to(np.asarray(nonlinear, np.float32), (n,)).copy()
        fb = np.broadcast_to(np.asarray(feedback, np.float32), (n,)).copy()
        self.lengths = np.asarray(lengths, np.int64)
        self.channels = ch
        self.jobs = (StreamJob * n)()
        in_off = out_off = 0
        self.in_offs, self.out_offs, self.out_caps, self.frame_offs, self.frames = [], [], [], [], []
        fo = 0
        for i in range(n):
            cap = plan.out_capacity(int(self.lengths[i]), float(sp[i]), float(nlv[i]))
            j = self.jobs[i]
            j.in_off, j.n_in, j.out_off, j.out_cap = in_off, int(self.lengths[i]), out_off, cap
            j.channels, j.speed, j.nonlinear, j.feedback = int(ch[i]), float(sp[i]), float(nlv[i]), float(fb[i])
            self.in_offs.append(in_off)
            self.out_offs.append(out_off)
            self.out_caps.append(cap)
            T = plan.frames(int(self.lengths[i])) if nlv[i] != 0 else 0
            self.frame_offs.append(fo)
            self.frames.append(T)
            fo += T
            in_off += int(self.lengths[i]) * int(ch[i])
            out_off += cap * int(ch[i])
        self.total_in, self.total_out, self.total_frames = in_off, out_off, fo
        dev = torch.device(device)
        self.device = dev
        self.d_in = torch.zeros(max(1, in_off) + 64, dtype=torch.int16, device=dev)
        self.d_out = torch.zeros(max(1, out_off), dtype=torch.int16, device=dev)
        self.d_nout = torch.zeros(n, dtype=torch.int64, device=dev)
        wsb = plan.L.spx_batch_workspace_bytes(plan.h, self.jobs, n)
        self.d_ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        self.taps = None
        if taps:
            T1 = max(1, fo)
            self.t_tension = torch.zeros(T1, dtype=torch.float32, device=dev)
            self.t_speed = torch.zeros(T1, dtype=torch.float32, device=dev)
            self.t_features = torch.zeros(T1 * 15, dtype=torch.float32, device=dev)
            self.taps = Taps(self.t_tension.data_ptr(), self.t_speed.data_ptr(), self.t_features.data_ptr(), None,
                             None)
            if spectrogram_taps:
                self.t_spec = torch.zeros(T1 * plan.N, dtype=torch.float32, device=dev)
                self.t_norm = torch.zeros(T1 * plan.W, dtype=torch.float32, device=dev)
                self.taps.spectrogram = self.t_spec.data_ptr()
                self.taps.normalized = self.t_norm.data_ptr()

    def upload(self, streams):
        """streams: list of int16 numpy arrays (interleaved).  Packs and copies them to HBM."""
        host = np.zeros(self.d_in.numel(), np.int16)
        for i, x in enumerate(streams):
            x = np.ascontiguousarray(x, np.int16).ravel()
            assert x.size == int(self.lengths[i]) * int(self.channels[i])
            host[self.in_offs[i]:self.in_offs[i] + x.size] = x
        self.d_in.copy_(torch.from_numpy(host))

    def run(self, stream=None):
        """Enqueue the whole hot path (analysis + walk) for the batch on `stream` (torch stream or None)."""
        hs = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        rc = self.plan.L.spx_batch_run(self.plan.h, self.jobs, self.n, self.d_in.data_ptr(), self.d_out.data_ptr(),
                                       self.d_nout.data_ptr(), self.d_ws.data_ptr(), self.d_ws.numel(),
                                       C.byref(self.taps) if self.taps is not None else None, hs)
        if rc != 0:
            raise RuntimeError("spx_batch_run: " + self.plan.L.spx_last_error().decode())

    def run_ahead(self, stream=None, in_ready=None, overlap=False):
        """spx_batch_run_ahead: like run(), software-pipelined with the previous call on the same stream (the caller alternates
        two Batch objects).  in_ready: a recorded torch.cuda.Event behind whatever completes the input (None: the input is
        resident when the call is made).  overlap: spx_batch_run_overlapped -- the walk kernels of consecutive calls overlap
        too; nothing enqueued between two calls may touch the later call's buffers (include/speedy_hip.h)."""
        hs = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        if overlap:
            assert in_ready is None
            rc = self.plan.L.spx_batch_run_overlapped(self.plan.h, self.jobs, self.n, self.d_in.data_ptr(), self.d_out.data_ptr(),
                                                      self.d_nout.data_ptr(), self.d_ws.data_ptr(), self.d_ws.numel(),
                                                      C.byref(self.taps) if self.taps is not None else None, hs)
            if rc != 0:
                raise RuntimeError("spx_batch_run_overlapped: " + self.plan.L.spx_last_error().decode())
            return
        rc = self.plan.L.spx_batch_run_ahead_when(self.plan.h, self.jobs, self.n, self.d_in.data_ptr(), self.d_out.data_ptr(),
                                                  self.d_nout.data_ptr(), self.d_ws.data_ptr(), self.d_ws.numel(),
                                                  C.byref(self.taps) if self.taps is not None else None, hs,
                                                  in_ready.cuda_event if in_ready is not None else None)
        if rc != 0:
            raise RuntimeError("spx_batch_run_ahead: " + self.plan.L.spx_last_error().decode())

    def step_counts(self):
        """Pitch searches per stream of the last run (the length of each stream's dependent chain); synchronises."""
        steps = (C.c_int32 * self.n)()
        rc = self.plan.L.spx_batch_read_steps(self.plan.h, self.jobs, self.n, self.d_ws.data_ptr(), steps,
                                              torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("spx_batch_read_steps: " + self.plan.L.spx_last_error().decode())
        torch.cuda.synchronize(self.device)
        return np.frombuffer(steps, np.int32).copy()

    def results(self):
        """Synchronise and return per-stream int16 outputs (host numpy)."""
        torch.cuda.synchronize(self.device)
        nout = self.d_nout.cpu().numpy()
        lost = nout == np.iinfo(np.int64).min   # SPX_NOUT_LOST_PRODUCER
        if lost.any():
            raise RuntimeError("a producer kernel never delivered its frames to streams %s (device-side poll limit)"
                               % np.nonzero(lost)[0][:8])
        if (nout < 0).any():
            raise RuntimeError("output capacity exceeded for streams %s" % np.nonzero(nout < 0)[0][:8])
        out = self.d_out.cpu().numpy()
        res = []
        for i in range(self.n):
            c = int(self.channels[i])
            res.append(out[self.out_offs[i]:self.out_offs[i] + int(nout[i]) * c].copy())
        return res

    def pack_outputs(self, stream=None):
        """Enqueue the gather of all produced frames into one contiguous device buffer (after run() on the same
        stream).  Returns (packed int16 device tensor, offsets int64[n+1] device tensor); stream i is
        packed[offsets[i]:offsets[i+1]].  One device-to-host copy then moves a whole batch's output."""
        if getattr(self, "d_packed", None) is None:
            self.d_packed = torch.empty_like(self.d_out)
            self.d_offsets = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
        hs = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        rc = self.plan.L.spx_batch_pack_outputs(self.jobs, self.n, self.d_out.data_ptr(), self.d_nout.data_ptr(),
                                                self.d_packed.data_ptr(), self.d_offsets.data_ptr(), hs)
        if rc != 0:
            raise RuntimeError("spx_batch_pack_outputs: " + self.plan.L.spx_last_error().decode())
        return self.d_packed, self.d_offsets

    def tap_arrays(self, i):
        """tension/speed/features rows of stream i (host numpy)."""
        torch.cuda.synchronize(self.device)
        fo, T = self.frame_offs[i], self.frames[i]
        K = max(0, T - self.plan.F + 1)
        r = dict(tension=self.t_tension[fo:fo + K].cpu().numpy(), speed=self.t_speed[fo:fo + K].cpu().numpy(),
                 features=self.t_features[fo * 15:(fo + K) * 15].cpu().numpy().reshape(K, 15))
        if getattr(self, "t_spec", None) is not None:
            N, W = self.plan.N, self.plan.W
            r["spectrogram"] = self.t_spec[fo * N:(fo + T) * N].cpu().numpy().reshape(T, N)
            r["normalized"] = self.t_norm[fo * W:(fo + K) * W].cpu().numpy().reshape(K, W)
        return r


class MixedBatch:
    """Streams of DIFFERENT sample rates in one call (spx_batch_run_mixed): stream i is served by plans[plan_index[i]].
    Inputs and outputs are packed like Batch's; results() returns the outputs in the order the streams were given."""

    def __init__(self, plans, plan_index, lengths, channels, speed, nonlinear=1.0, feedback=0.0, device="cuda", taps=False):
        n = len(lengths)
        self.plans, self.n = list(plans), n
        self.L = plans[0].L
        self.pidx = [int(v) for v in plan_index]
        self.plan_index = (C.c_int * n)(*self.pidx)
        self.hplans = (C.c_void_p * len(plans))(*[p.h for p in plans])
        ch = np.broadcast_to(np.asarray(channels, np.int32), (n,)).copy()
        sp = np.broadcast_to(np.asarray(speed, np.float32), (n,)).copy()
        nlv = np.broadcast_to(np.asarray(nonlinear, np.float32), (n,)).copy()
        fb = np.broadcast_to(np.asarray(feedback, np.float32), (n,)).copy()
        self.lengths, self.channels = np.asarray(lengths, np.int64), ch
        self.jobs = (StreamJob * n)()
        in_off = out_off = 0
        self.in_offs, self.out_offs, self.frames = [], [], []
        for i in range(n):
            pl = plans[self.pidx[i]]
            cap = pl.out_capacity(int(self.lengths[i]), float(sp[i]), float(nlv[i]))
            j = self.jobs[i]
            j.in_off, j.n_in, j.out_off, j.out_cap = in_off, int(self.lengths[i]), out_off, cap
            j.channels, j.speed, j.nonlinear, j.feedback = int(ch[i]), float(sp[i]), float(nlv[i]), float(fb[i])
            self.in_offs.append(in_off)
            self.out_offs.append(out_off)
            self.frames.append(pl.frames(int(self.lengths[i])) if nlv[i] != 0 else 0)
            in_off += int(self.lengths[i]) * int(ch[i])
            out_off += cap * int(ch[i])
        dev = torch.device(device)
        self.device = dev
        self.total_frames_in = int(self.lengths.sum())
        self.d_in = torch.zeros(max(1, in_off) + 64, dtype=torch.int16, device=dev)
        self.d_out = torch.zeros(max(1, out_off), dtype=torch.int16, device=dev)
        self.d_nout = torch.zeros(n, dtype=torch.int64, device=dev)
        wsb = self.L.spx_batch_workspace_bytes_mixed(self.hplans, len(plans), self.jobs, self.plan_index, n)
        if wsb == 0:
            raise RuntimeError("spx_batch_workspace_bytes_mixed: " + self.L.spx_last_error().decode())
        self.d_ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        self.taps = None
        if taps:
            # tap rows: streams grouped by plan index, job order inside a group (include/speedy_hip.h)
            self.tap_off = [0] * n
            rows = 0
            for g in range(len(plans)):
                for i in range(n):
                    if self.pidx[i] == g:
                        self.tap_off[i] = rows
                        rows += self.frames[i]
            T1 = max(1, rows)
            self.t_tension = torch.zeros(T1, dtype=torch.float32, device=dev)
            self.t_speed = torch.zeros(T1, dtype=torch.float32, device=dev)
            self.t_features = torch.zeros(T1 * 15, dtype=torch.float32, device=dev)
            self.taps = Taps(self.t_tension.data_ptr(), self.t_speed.data_ptr(), self.t_features.data_ptr(), None, None)

    def upload(self, streams):
        host = np.zeros(self.d_in.numel(), np.int16)
        for i, x in enumerate(streams):
            x = np.ascontiguousarray(x, np.int16).ravel()
            assert x.size == int(self.lengths[i]) * int(self.channels[i])
            host[self.in_offs[i]:self.in_offs[i] + x.size] = x
        self.d_in.copy_(torch.from_numpy(host))

    def run(self, stream=None):
        hs = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        rc = self.L.spx_batch_run_mixed_taps(self.hplans, len(self.plans), self.jobs, self.plan_index, self.n,
                                             self.d_in.data_ptr(), self.d_out.data_ptr(), self.d_nout.data_ptr(),
                                             self.d_ws.data_ptr(), self.d_ws.numel(),
                                             C.byref(self.taps) if self.taps is not None else None, hs)
        if rc != 0:
            raise RuntimeError("spx_batch_run_mixed: " + self.L.spx_last_error().decode())

    def run_ahead(self, stream=None):
        """spx_batch_run_mixed_ahead: like run() (no taps), software-pipelined with the previous call on the same stream (the
        caller alternates two MixedBatch objects over the same plans; inputs are resident when the call is made)."""
        assert self.taps is None
        hs = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        rc = self.L.spx_batch_run_mixed_ahead(self.hplans, len(self.plans), self.jobs, self.plan_index, self.n,
                                              self.d_in.data_ptr(), self.d_out.data_ptr(), self.d_nout.data_ptr(),
                                              self.d_ws.data_ptr(), self.d_ws.numel(), hs)
        if rc != 0:
            raise RuntimeError("spx_batch_run_mixed_ahead: " + self.L.spx_last_error().decode())

    def step_counts(self):
        steps = (C.c_int32 * self.n)()
        torch.cuda.synchronize(self.device)
        rc = self.L.spx_batch_read_steps_mixed(self.hplans, len(self.plans), self.jobs, self.plan_index, self.n,
                                               self.d_ws.data_ptr(), steps, torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("spx_batch_read_steps_mixed: " + self.L.spx_last_error().decode())
        return np.frombuffer(steps, np.int32).copy()

    def counts(self):
        """Synchronise; produced frames per stream (raises on overflow / a lost producer)."""
        torch.cuda.synchronize(self.device)
        nout = self.d_nout.cpu().numpy()
        if (nout < 0).any():
            raise RuntimeError("output capacity exceeded / lost producer for streams %s" % np.nonzero(nout < 0)[0][:8])
        return nout

    def results(self):
        nout = self.counts()
        out = self.d_out.cpu().numpy()
        return [out[self.out_offs[i]:self.out_offs[i] + int(nout[i]) * int(self.channels[i])].copy() for i in range(self.n)]

    def crcs(self):
        """CRC-32 of every stream's output bytes (what tools/check_scale.py and bench.py compare)."""
        import zlib
        nout = self.counts()
        out = self.d_out.cpu().numpy()
        return [zlib.crc32(out[self.out_offs[i]:self.out_offs[i] + int(nout[i]) * int(self.channels[i])].tobytes())
                for i in range(self.n)]

    def tap_arrays(self, i):
        torch.cuda.synchronize(self.device)
        fo, T = self.tap_off[i], self.frames[i]
        K = max(0, T - self.plans[self.pidx[i]].F + 1)
        return dict(tension=self.t_tension[fo:fo + K].cpu().numpy(), speed=self.t_speed[fo:fo + K].cpu().numpy(),
                    features=self.t_features[fo * 15:(fo + K) * 15].cpu().numpy().reshape(K, 15))


class Pipeline:
    """spx_pipeline (include/speedy_hip.h): batch after batch of ONE shape, host memory to host memory -- the library owns the
    device buffer sets, the pinned output buffers, its streams and events, and overlaps copies in, kernels and copies out of up
    to `depth` batches.  The caller loop of the reference (speedy_wave.cc:154-242: write a chunk, read what is ready) for a caller
    whose unit is a batch of streams.

        pipe = Pipeline(plan, lengths, channels, speed)
        t = pipe.submit(x)            # x: packed int16 input of one batch (numpy / pinned torch tensor; device tensor: device=True)
        outs = pipe.results(t)        # list of per-stream int16 arrays (copies); pipe.wait(t) returns the raw views

    plans / plan_index: a batch that mixes sample rates (spx_pipeline_create_mixed)."""

    def __init__(self, plan, lengths, channels, speed, nonlinear=1.0, feedback=0.0, depth=0, device_out=False, plan_index=None):
        plans = list(plan) if isinstance(plan, (list, tuple)) else [plan]
        self.plans = plans
        self.L = plans[0].L
        n = len(lengths)
        self.n = n
        ch = np.broadcast_to(np.asarray(channels, np.int32), (n,)).copy()
        sp = np.broadcast_to(np.asarray(speed, np.float32), (n,)).copy()
        nlv = np.broadcast_to(np.asarray(nonlinear, np.float32), (n,)).copy()
        fb = np.broadcast_to(np.asarray(feedback, np.float32), (n,)).copy()
        self.lengths, self.channels = np.asarray(lengths, np.int64), ch
        self.jobs = (StreamJob * n)()
        self.in_offs = []
        in_off = 0
        for i in range(n):
            j = self.jobs[i]
            j.in_off, j.n_in, j.out_off, j.out_cap = in_off, int(self.lengths[i]), 0, 0   # (the pipeline lays the outputs out itself)
            j.channels, j.speed, j.nonlinear, j.feedback = int(ch[i]), float(sp[i]), float(nlv[i]), float(fb[i])
            self.in_offs.append(in_off)
            in_off += int(self.lengths[i]) * int(ch[i])
        self.total_in = in_off
        self.device_out = bool(device_out)
        flags = 1 if device_out else 0
        if plan_index is None and len(plans) == 1:
            self.h = self.L.spx_pipeline_create(plans[0].h, self.jobs, n, int(depth), flags)
        else:
            self._pidx = (C.c_int * n)(*[int(v) for v in (plan_index if plan_index is not None else [0] * n)])
            self._hplans = (C.c_void_p * len(plans))(*[p.h for p in plans])
            self.h = self.L.spx_pipeline_create_mixed(self._hplans, len(plans), self.jobs, self._pidx, n, int(depth), flags)
        if not self.h:
            raise RuntimeError("spx_pipeline_create: " + self.L.spx_last_error().decode())
        self.depth = self.L.spx_pipeline_depth(self.h)
        assert self.L.spx_pipeline_input_values(self.h) == self.total_in or n == 0
        self._keep = {}   # ticket -> the input object (host memory must stay alive until its copy has been made)

    def pack(self, streams):
        """One batch's input as the pipeline expects it: the streams (int16 numpy arrays, interleaved) one after the other."""
        host = np.zeros(self.total_in, np.int16)
        for i, x in enumerate(streams):
            x = np.ascontiguousarray(x, np.int16).ravel()
            assert x.size == int(self.lengths[i]) * int(self.channels[i])
            host[self.in_offs[i]:self.in_offs[i] + x.size] = x
        return host

    def host_input(self):
        """The pinned staging buffer of the NEXT submit as an int16 numpy view (fill it, then submit(it))."""
        ptr = self.L.spx_pipeline_host_input(self.h)
        if not ptr:
            raise RuntimeError("spx_pipeline_host_input: " + self.L.spx_last_error().decode())
        return np.ctypeslib.as_array((C.c_int16 * self.total_in).from_address(ptr))

    def submit(self, x, device=False):
        if isinstance(x, torch.Tensor):
            assert x.dtype == torch.int16 and x.is_contiguous() and x.numel() >= self.total_in
            device = x.is_cuda
            # (include/speedy_hip.h: a device input is used in place and must be allocated 64 values past the last stream's end)
            assert not device or x.numel() >= self.total_in + 64, "device input: allocate spx_pipeline_input_values() + 64 int16 values"
            ptr = x.data_ptr()
        else:
            x = np.ascontiguousarray(x, np.int16)
            assert x.size >= self.total_in
            ptr = x.ctypes.data
        t = self.L.spx_pipeline_submit(self.h, ptr, 1 if device else 0)
        if t < 0:
            raise RuntimeError("spx_pipeline_submit: " + self.L.spx_last_error().decode())
        self._keep[t] = x
        self._keep.pop(t - 2 * self.depth, None)
        return t

    def input_consumed(self, ticket):
        """Blocks until the input handed over with `ticket` may be overwritten (host input: copied in; device input: batch done)."""
        if self.L.spx_pipeline_input_consumed(self.h, int(ticket)) != 0:
            raise RuntimeError("spx_pipeline_input_consumed: " + self.L.spx_last_error().decode())

    def wait(self, ticket):
        """(out, offsets, counts) of a batch: numpy views of the pipeline's pinned host buffers -- device_out: (data pointer of the
        int16 output in device memory, offsets as a numpy array, data pointer of the int64 counts in device memory)."""
        o, f, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self.L.spx_pipeline_wait(self.h, int(ticket), C.byref(o), C.byref(f), C.byref(c))
        if rc != 0:
            raise RuntimeError("spx_pipeline_wait: " + self.L.spx_last_error().decode())
        offsets = np.ctypeslib.as_array((C.c_int64 * (self.n + 1)).from_address(f.value))
        if self.device_out:
            return o.value, offsets, c.value
        counts = np.ctypeslib.as_array((C.c_int64 * self.n).from_address(c.value))
        out = np.ctypeslib.as_array((C.c_int16 * max(1, int(offsets[self.n]))).from_address(o.value))
        return out, offsets, counts

    def results(self, ticket):
        """Per-stream int16 outputs of a batch (copies; raises on overflow / a lost producer)."""
        out, offsets, counts = self.wait(ticket)
        if self.device_out:
            cnt = torch.empty(self.n, dtype=torch.int64)
            self.L.spx_copy_to_host(cnt.data_ptr(), counts, self.n * 8, None)
            self.L.spx_stream_synchronize(None)
            counts = cnt.numpy()
            total = int(offsets[self.n])
            host = torch.empty(total, dtype=torch.int16)
            self.L.spx_copy_to_host(host.data_ptr(), out, total * 2, None)
            self.L.spx_stream_synchronize(None)
            out = host.numpy()
        if (counts < 0).any():
            raise RuntimeError("output capacity exceeded / lost producer for streams %s" % np.nonzero(counts < 0)[0][:8])
        return [out[int(offsets[i]):int(offsets[i]) + int(counts[i]) * int(self.channels[i])].copy() for i in range(self.n)]

    def close(self):
        if self.h:
            self.L.spx_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def compress_batch(streams, sample_rate, channels, speed, nonlinear=1.0, feedback=0.0, match_matlab=False,
                   taps=False, spectrogram_taps=False):
    """One-call convenience: returns (list of outputs, Batch)."""
    plan = Plan(sample_rate, match_matlab)
    ch = np.broadcast_to(np.asarray(channels, np.int32), (len(streams),))
    lengths = [np.asarray(x).size // int(c) for x, c in zip(streams, ch)]
    b = Batch(plan, lengths, channels, speed, nonlinear, feedback, taps=taps, spectrogram_taps=spectrogram_taps)
    b.upload(streams)
    b.run()
    return b.results(), b
