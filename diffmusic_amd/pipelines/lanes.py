"""Clip lanes: the denoising loop of ONE GPU software-pipelined over groups of clips.

The reference loop body (diffmusic/pipelines/pipeline_musicldm.py:690-758) is U-Net -> CFG -> scheduler.step for the whole
batch, strictly in that order.  On the MI355X the two halves behave differently: the U-Net is a chain of several hundred small
launches that leave most of the 256 CUs idle (latency-bound), the guidance sweep inside `scheduler.step` (VAE decode, HiFi-GAN,
mel loss and the backward pass through all of them) fills every CU.  Clips are independent (per-clip norms, per-clip
generators), so the batch is cut into lanes and the loop is run staggered:

    sweep stream (normal priority) :  sweep A(i) | sweep B(i) | sweep A(i+1) | sweep B(i+1) | ...
    U-Net stream (high priority)   :  .......... | unet A(i+1) | unet B(i+1) | unet A(i+2)  | ...

Lane A's next U-Net forward depends only on lane A's finished step, so it runs UNDER lane B's sweep, its workgroups taking the
compute units the sweep's tiles release (stream priority decides who gets a freed CU, nothing is pre-empted).  The network
executors and their workspaces are shared by the lanes: every U-Net forward is on the one U-Net stream and every sweep on the
one sweep stream, so no two launches that use the same workspace can overlap.

Per clip the arithmetic is the one of a batch of the lane's size: a lane is exactly `Pipeline.__call__` on its clips (same
generators, conditioning rows and measurement rows), which `tests/test_gpu_lanes.py` checks bit for bit."""
import os

import torch


def split_sizes(n, lanes):
    """Contiguous lane sizes: n clips over at most `lanes` lanes, the first lanes one clip larger when n % lanes != 0."""
    lanes = max(1, min(int(lanes), n))
    q, r = divmod(n, lanes)
    return [q + (1 if k < r else 0) for k in range(lanes)]


class Lane:
    """State of one clip group.  `ids` are its clips' positions in the call's batch."""

    def __init__(self, ids, latents, cond, measurement, generator):
        self.ids, self.latents, self.cond, self.measurement, self.generator = ids, latents, cond, measurement, generator
        self.ev_unet = self.ev_sweep = None
        self.eps = None
        self.losses = []          # one device tensor per finished step
        self.pending = []         # losses not yet covered by a NaN check
        self.nan_events = {}      # step -> event recorded behind the NaN flag's copy to the host


class LaneRunner:
    """Two streams + the staggered enqueue order.  `unet_fn(lane, i)` and `step_fn(lane, i, eps)` enqueue on torch's CURRENT
    stream (the runner makes the right one current) and return eps / (prev_sample, loss)."""

    def __init__(self, device, unet_priority=None):
        self.device = torch.device(device)
        # host tensors (the CPU stand-in engines of tests/stubs.py): the same enqueue order and NaN protocol without streams -- only the
        # host logic is exercised there, the product runs on the GPU
        self.on_gpu = self.device.type == "cuda"
        self.U = self.S = None
        self._flags = None
        if not self.on_gpu:
            return
        if unet_priority is None:                         # DMX_LANE_UNET_PRIORITY=0: both streams of equal priority (A/B measurements)
            unet_priority = int(os.environ.get("DMX_LANE_UNET_PRIORITY", "-1"))
        with torch.cuda.device(self.device):
            mask = os.environ.get("DMX_LANE_CU_MASK")     # experiment: hex word, repeated over the 8 mask words = the U-Net stream's CUs; the
            if mask:                                      # sweep stream gets the complement (hipExtStreamCreateWithCUMask)
                self.U, self.S = self._masked_streams(int(mask, 16))
            else:
                # lower number = higher priority; torch clamps to the device's range
                self.U = torch.cuda.Stream(device=self.device, priority=unet_priority)
                self.S = torch.cuda.Stream(device=self.device, priority=0)

    @staticmethod
    def _masked_streams(word):
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        out = []
        for w in (word & 0xffffffff, ~word & 0xffffffff):
            s = C.c_void_p()
            arr = (C.c_uint32 * 8)(*([w] * 8))
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), arr)
            if rc != 0:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
            out.append(torch.cuda.ExternalStream(s.value))
        return out

    def _enqueue_unet(self, lane, i, unet_fn):
        if not self.on_gpu:
            lane.eps = unet_fn(lane, i)
            return
        if lane.ev_sweep is not None:
            self.U.wait_event(lane.ev_sweep)              # this lane's latents of step i - 1
        with torch.cuda.stream(self.U):
            lane.eps = unet_fn(lane, i)
            lane.ev_unet = torch.cuda.Event()
            lane.ev_unet.record(self.U)
        lane.eps.record_stream(self.S)                    # allocated on U, read by the sweep on S

    def _enqueue_sweep(self, lane, i, step_fn, want_flag):
        if not self.on_gpu:
            prev, loss = step_fn(lane, i, lane.eps)
            lane.eps, lane.latents = None, prev
            lane.losses.append(loss)
            lane.pending.append(loss)
            if want_flag:
                self._flags[lane.index, i] = bool(torch.isnan(torch.cat([l.float().reshape(-1) for l in lane.pending])).any())
                lane.pending = []
                lane.nan_events[i] = None
            return
        self.S.wait_event(lane.ev_unet)
        with torch.cuda.stream(self.S):
            prev, loss = step_fn(lane, i, lane.eps)
            lane.eps = None
            lane.latents = prev
            lane.ev_sweep = torch.cuda.Event()
            lane.ev_sweep.record(self.S)
            lane.losses.append(loss)
            lane.pending.append(loss)
            if want_flag:                                 # NaN in any loss since the last check -> one byte to pinned host memory
                k = lane.index
                self._flags_dev[k, i] = torch.isnan(torch.cat([l.float().reshape(-1) for l in lane.pending])).any()
                lane.pending = []
                self._flags[k, i:i + 1].copy_(self._flags_dev[k, i:i + 1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.S)
                lane.nan_events[i] = ev
        prev.record_stream(self.U)                        # allocated on S, read by the next U-Net forward on U

    def run(self, lanes, n_steps, unet_fn, step_fn, nan_check_every=1, on_step=None):
        """Runs all `n_steps` for every lane.  Returns the index of the first step at which a NaN loss was seen by a host check
        (None: the trajectory finished).  NaN checks happen every `nan_check_every` steps and at the last step, per lane, one
        lane-step behind the enqueue front so that the GPU always has the other lane's sweep to run while the host waits."""
        for k, ln in enumerate(lanes):
            ln.index = k
            ln.ev_unet = ln.ev_sweep = None
            ln.losses, ln.pending, ln.nan_events = [], [], {}
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.device)
            self.U.wait_stream(cur)
            self.S.wait_stream(cur)
            for ln in lanes:
                ln.latents.record_stream(self.U)
                ln.latents.record_stream(self.S)
            self._flags = torch.zeros(len(lanes), n_steps, dtype=torch.bool).pin_memory()
            self._flags_dev = torch.zeros(len(lanes), n_steps, dtype=torch.bool, device=self.device)
            self.S.wait_stream(cur)                       # (the flag buffer's fill above)
        else:
            self._flags = torch.zeros(len(lanes), n_steps, dtype=torch.bool)
        every = max(1, int(nan_check_every))

        def due(i):
            return (i + 1) % every == 0 or i == n_steps - 1

        bad_at = None
        try:
            for ln in lanes:                              # prologue: step 0 of every lane
                self._enqueue_unet(ln, 0, unet_fn)
                self._enqueue_sweep(ln, 0, step_fn, due(0))
            for i in range(n_steps):
                for ln in lanes:
                    if due(i):
                        ev = ln.nan_events.pop(i)
                        if ev is not None:
                            ev.synchronize()
                        if bool(self._flags[ln.index, i]):
                            bad_at = i
                            return bad_at
                    if i + 1 < n_steps:
                        self._enqueue_unet(ln, i + 1, unet_fn)
                        self._enqueue_sweep(ln, i + 1, step_fn, due(i + 1))
                if on_step is not None:
                    on_step(i)
            return None
        finally:
            if self.on_gpu:
                if bad_at is not None:                    # a restart follows: let the work already enqueued drain first
                    self.S.synchronize()
                    self.U.synchronize()
                cur.wait_stream(self.S)
                cur.wait_stream(self.U)
                for ln in lanes:
                    ln.latents.record_stream(cur)
