"""Sampling pipelines: drop-in for pl_trainer/inference/inference.py (same class names, constructor
and call signatures, same returned dict keys).

  Inference.__init__                                   inference.py:26-51
  InferenceIP2PVideo.__call__                          inference.py:162-218
  InferenceIP2PVideo.second_clip_forward               inference.py:220-289
  InferenceIP2PVideoOpticalFlow.second_clip_forward    inference.py:313-398

Per DDIM step the host issues: one input-assembly kernel, one hipGraph replay of the whole UNet
forward (the 3 CFG branches run as three concurrent kernel chains on three HIP streams captured in
ONE graph, so their latency-bound kernels overlap and fill each other's tails: +10 % over batching
them in every launch; text K/V hoisted out of the loop), and one fused CFG-combine +
noise-correction + scheduler-step kernel.  Latents stay fp32 in the reference
layout [1,F,4,h,w]; there is no host<->device traffic inside the loop.
"""
import os
import weakref

import torch

from . import ops
from .schedulers import DDIMScheduler, DDPMScheduler
from .flow_utils import resize_flow


class GraphedUNet:
    """One captured hipGraph of UNet3DConditionModel.forward_cl for fixed (B,F,H,W,ctx_len).

    Static inputs: x_in (channels-last fp16), t (device fp32), per-layer text K/V; static output eps.
    """

    def __init__(self, unet, B, F, H, W, ctx_len, use_graph=True, branch_streams=False, cfg_clips=0):
        dev = unet.device
        self.branch_streams = branch_streams and B > 1
        # cfg_clips = n: the B = 3 n samples are the CFG branches of n clips, BRANCH-major, with the inputs of branches 1 and 2 identical
        # (what build_unet_input writes): UNet3DConditionModel.forward_cl computes their common prefix once
        self.cfg_clips = cfg_clips if (cfg_clips > 0 and B == 3 * cfg_clips and not self.branch_streams) else 0
        self._streams = None
        # weak: the process-wide graph cache (shared_runner) must not keep a UNet - 2.5 GB of weights - alive
        try:
            self._unet = weakref.ref(unet)
        except TypeError:
            self._unet = lambda u=unet: u
        self.key = (B, F, H, W, ctx_len)
        self.x_in = torch.zeros((B * F * H * W, unet.in_pad), device=dev, dtype=torch.float16)
        self.t = torch.zeros((B,), device=dev, dtype=torch.float32)
        self.kvs = None   # per-layer text K/V (tensor, or (K/V, fragment streams) where the fused cross-attention runs): set_context
        self.use_graph = use_graph
        self.graph = None
        self.eps = None
        self.start = 0
        # split-K scratch owned by this runner (one per concurrently running branch), allocated outside any capture
        self._ws = [ops.new_workspace(dev) for _ in range(B if self.branch_streams else 1)]

    @property
    def unet(self):
        u = self._unet()
        if u is None:
            raise RuntimeError("GraphedUNet: its UNet has been garbage collected")
        return u

    def set_context(self, ctx):
        kvs, L = self.unet.project_context(ctx)
        assert L == self.key[4] and ctx.shape[0] == self.key[0]
        if self.kvs is None:   # static buffers of the captured graph, shaped like the first projection
            self.kvs = [tuple(torch.zeros_like(t) for t in kv) if isinstance(kv, tuple) else torch.zeros_like(kv) for kv in kvs]
        for dst, src in zip(self.kvs, kvs):
            for d, s_ in zip(dst, src) if isinstance(dst, tuple) else ((dst, src),):
                d.copy_(s_)

    def _forward(self):
        B, F, H, W, L = self.key
        if not self.branch_streams:
            with ops.workspace(self._ws[0]):
                return self.unet.forward_cl(self.x_in, self.t, self.kvs, L, B, F, H, W, start=self.start, cfg_clips=self.cfg_clips)
        # one HIP stream per CFG branch: the branches are independent, so their (latency-bound) kernels
        # overlap and fill each other's tails; fork/join is captured into the same hipGraph.
        if self._streams is None:
            self._streams = [torch.cuda.Stream() for _ in range(B)]
        main = torch.cuda.current_stream()
        rows = F * H * W
        outs = [None] * B
        for b, st in enumerate(self._streams):
            st.wait_stream(main)
            with torch.cuda.stream(st), ops.workspace(self._ws[b]):
                kvs = [(kv[0][b * L:(b + 1) * L], kv[1][b:b + 1]) if isinstance(kv, tuple) else kv[b * L:(b + 1) * L] for kv in self.kvs]
                outs[b] = self.unet.forward_cl(self.x_in[b * rows:(b + 1) * rows], self.t[b:b + 1], kvs, L, 1, F, H, W,
                                               start=self.start)
        for st in self._streams:
            main.wait_stream(st)
        return torch.cat(outs, 0)

    def run(self):
        """eps [B*F*H*W, 4] fp32 for the current contents of x_in / t / kvs."""
        if not self.use_graph:
            return self._forward()
        if self.graph is None:
            # warm-up on a side stream (sets lazy kernel attributes, primes the allocator), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._forward()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.eps = self._forward()
            self.graph = g
        self.graph.replay()
        return self.eps


class Inference:
    def __init__(self, unet, scheduler="ddim", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 num_ddim_steps=20, guidance_scale=5, use_graph=True, branch_streams=True):
        self.unet = unet
        if scheduler == "ddim":
            cls, kw = DDIMScheduler, {"set_alpha_to_one": False, "steps_offset": 1, "clip_sample": False}
        elif scheduler == "ddpm":
            cls, kw = DDPMScheduler, {"clip_sample": False}
        else:
            raise NotImplementedError()
        self.scheduler = cls(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        self.scheduler.set_timesteps(num_ddim_steps)
        self.num_ddim_steps = num_ddim_steps
        self.guidance_scale = guidance_scale
        self.use_graph = use_graph
        self.branch_streams = branch_streams
        self.variance_noises = None  # optional injected DDPM noises, one [1,F,4,h,w] tensor (or None) per step

    def _runner(self, B, F, H, W, L, slot=0):
        return shared_runner(self.unet, B, F, H, W, L, slot, self.use_graph, self.branch_streams)


# Threading: ONE host thread drives the pipelines of a process (like the reference): runners share static input / output
# buffers, and ops._WS_OVERRIDE (the split-K scratch selection) is a plain module global.
# Captured UNet graphs are shared process-wide, keyed by (UNet object, its weights version, shape, slot, mode): pipe
# objects are cheap and short-lived (one per clip / scheduler setting in the drivers and tests), the captured hipGraph of a
# full-size 3-stream forward is not - and capturing the same shape again right after destroying a graph crashed inside
# hipGraphLaunch on ROCm 7.2 (first replay of the new graph; reproducible with tests/test_full_size_gpu.py followed by
# tests/test_model_gpu.py, also on the round-1 code).  Entries die with their UNet or when its weights are reloaded:
# a captured graph holds the OLD weight pointers.
_RUNNERS = {}
_FINALIZERS = {}
_DEAD = set()   # UNets collected while a stream capture was in progress: their runners are purged at the next shared_runner() call


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing()


# hipGraphLaunch of a NEW graph segfaults on ROCm 7.2 once a captured graph of this process has been destroyed and the same shape is
# captured again (tests/test_model_gpu.py followed by tests/test_full_size_gpu.py: crash with destruction at collection time, at the next
# shared_runner() call, and with an empty_cache() in between; no crash when the dead graphs are merely parked - gpurun_out of round 3,
# tools/r03_run18.sh).  Retired graphs are therefore PARKED here, never destroyed (their static buffers and split-K scratch go with the
# runner; the graph's private activation pool - a few GB for a 3-branch full-size forward - stays until the process ends or
# release_dead_graphs() is called).  INSV2V_GRAPH_PURGE=destroy restores immediate destruction.
_GRAVEYARD = []
_PURGE_MODE = os.environ.get("INSV2V_GRAPH_PURGE", "keep")   # keep | destroy
# The park is bounded: beyond INSV2V_GRAPH_PARK_MAX graphs the OLDEST is destroyed behind a device synchronize.  A process that
# keeps ONE UNet never parks anything but the graphs of reloaded weights / retired stack sizes; the bound only bites in processes that
# build many models (test suites), where the oldest graphs belong to models collected long ago.
_PARK_MAX = int(os.environ.get("INSV2V_GRAPH_PARK_MAX", "32"))


def parked_graphs():
    """Number of retired hipGraphs currently parked (each holds its private activation pool)."""
    return len(_GRAVEYARD)


def _park(graph):
    _GRAVEYARD.append(graph)
    if len(_GRAVEYARD) > _PARK_MAX:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
        while len(_GRAVEYARD) > _PARK_MAX:
            _GRAVEYARD.pop(0)


def release_dead_graphs():
    """Destroy the parked graphs of retired runners (frees their activation pools).  Safe when no further graph will be captured in
    this process, or on a ROCm without the hipGraphLaunch crash described above."""
    if _GRAVEYARD and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
    _GRAVEYARD.clear()


def _purge_runners(uid, keep_version=None):
    dead = [k for k in _RUNNERS if k[0] == uid and k[1] != keep_version]
    if dead:
        gpu = torch.cuda.is_available() and torch.cuda.is_initialized()
        if gpu:
            torch.cuda.synchronize()  # never destroy a graph that may still be executing
        for k in dead:
            r = _RUNNERS.pop(k)
            if _PURGE_MODE == "keep" and getattr(r, "graph", None) is not None:
                _park(r.graph)
            del r


def _unet_collected(uid):
    """weakref.finalize callback of a UNet.  Its graphs are destroyed right away - EXCEPT when the collector happens to run inside a
    stream capture, where the device synchronize of _purge_runners is illegal: then the id is parked in _DEAD for the next
    shared_runner() call.  (Destroying them only there as a rule - i.e. immediately before the next capture of the same shapes - ran
    into the hipGraphLaunch crash described above: tests/test_model_gpu.py followed by tests/test_full_size_gpu.py.)"""
    _FINALIZERS.pop(uid, None)
    if _capturing():
        _DEAD.add(uid)
    else:
        _purge_runners(uid)


def shared_runner(unet, B, F, H, W, L, slot=0, use_graph=True, branch_streams=True, cfg_clips=0):
    """Process-wide runner of (unet, shape, slot).  One host thread drives a process's pipes (the runners' static buffers, ops' workspace
    override and this cache are plain module state)."""
    uid, ver = id(unet), getattr(unet, "weights_version", 0)
    while _DEAD:   # (an id can be reused by a NEW UNet: stale entries go first, so it never sees another model's graphs)
        _purge_runners(_DEAD.pop())
    if uid not in _FINALIZERS:
        try:
            fin = weakref.finalize(unet, _unet_collected, uid)
            fin.atexit = False  # at interpreter exit the HIP runtime may already be gone
            _FINALIZERS[uid] = fin
        except TypeError:  # not weak-referenceable (test doubles): entries live as long as the process
            _FINALIZERS[uid] = None
    _purge_runners(uid, keep_version=ver)
    key = (uid, ver, B, F, H, W, L, slot, bool(use_graph), bool(branch_streams), int(cfg_clips))
    r = _RUNNERS.get(key)
    if r is None:
        r = _RUNNERS[key] = GraphedUNet(unet, B, F, H, W, L, use_graph, branch_streams, cfg_clips)
    return r


MAX_CLIPS_IN_FLIGHT = int(os.environ.get("INSV2V_MAX_CLIPS", "20"))


def max_clips_in_flight(frames=16, h=32, w=48):
    """Clips that may be stacked into one launch chain: at most 20 (5 -> 10 clips gave +1.4 %, 10 -> 20 fills the last round of the
    level-0 tile grids - 22.5 -> 23 rounds instead of 11.25 -> 12 - and halves the launches per clip: +0.9 %, profiles/r04_clips20.txt), and few enough that every
    [3 * clips * F * h * w, 640] fp16 tensor (the input of the level-0 convolutions, the GEGLU rows of level 1) fits ONE 2 GiB buffer
    descriptor: the row kernels of the feed-forward / temporal / text attention blocks address whole operands through one - 20 for C2
    (B = 60, 1.89 GB).  Wider operands ([.., 960] fused q/k/v rows, [.., 960 / 1920] concatenations) go through kernels that rebase
    their descriptors (insv2v_rowlin, insv2v_attention) or through insv2v_gemm, which since round 5 runs any problem beyond the window
    as row / image ranges - so clips of more than 16 frames (unfused temporal attention, q/k/v rows from insv2v_gemm) stack by the
    same rule: 7 for C5 (round 4: 5)."""
    return max(1, min(MAX_CLIPS_IN_FLIGHT, (2 ** 31 - 2 ** 20) // (3 * frames * h * w * 640 * 2)))


class InferenceIP2PVideo(Inference):
    def zeros(self, x):
        return torch.zeros_like(x)

    # ---- shared loop ------------------------------------------------------------------------------
    def _clip_tensors(self, latent, img_cond):
        dev = self.unet.device
        if latent.shape[0] != 1:
            raise ValueError("internal: one clip per entry (a batch is split into clips by the callers)")
        lat = latent[0].to(device=dev, dtype=torch.float32).contiguous()
        cond = img_cond[0].to(device=dev, dtype=torch.float32).contiguous()
        if lat.shape != cond.shape or lat.shape[1] != 4:
            raise ValueError(f"latent {tuple(latent.shape)} and img_cond {tuple(img_cond.shape)} must both be [1,F,4,h,w]")
        return lat, cond

    def _latent_flows(self, flows, n_query, h, w):
        """Image-resolution flows (one [R,2,H,W] per query frame) -> [n_query,R,2,h,w] at latent resolution, resized ONCE per window
        (the reference repeats this loop-invariant resize every corrected step, inference.py:374-378)."""
        if len(flows) != n_query:
            raise ValueError("need one [R,2,H,W] flow per query frame")
        dev = self.unet.device
        return torch.stack([resize_flow(f.to(device=dev, dtype=torch.float32), (h, w)) for f in flows], 0).contiguous()

    def _prep(self, latent, text_cond, text_uncond, img_cond, slot=0):
        lat, cond = self._clip_tensors(latent, img_cond)
        ctx = torch.cat([text_uncond, text_uncond, text_cond], dim=0)
        F, _, h, w = lat.shape
        runner = self._runner(3, F, h, w, ctx.shape[1], slot)
        runner.set_context(ctx)
        return lat, cond, runner

    def _loop(self, *args, **kwargs):
        gen = self._loop_gen(*args, **kwargs)
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    def _loop_gen(self, latent, text_cond, text_uncond, img_cond, text_cfg, img_cfg, start_time, guidance_rescale,
                  latent_ref=None, noise_correct_step=0.0, flows=None, slot=0):
        """One sampling loop as a generator that yields after enqueueing each step's (asynchronous) GPU work,
        so several independent clips can be interleaved from one host thread (``run_concurrent``)."""
        lat, cond, runner = self._prep(latent, text_cond, text_uncond, img_cond, slot)
        dev = lat.device
        F, _, h, w = lat.shape
        ref = None
        if latent_ref is not None:
            ref = latent_ref[0].to(device=dev, dtype=torch.float32).contiguous()
        stats = torch.empty(2, device=dev, dtype=torch.float32) if guidance_rescale > 0 else None
        all_latent, all_pred = [], []
        for i, t in enumerate(self.scheduler.timesteps[start_time:]):
            t = int(t)
            ops.build_unet_input(lat, cond, runner.x_in, runner.t, t, 3)
            eps = runner.run()
            lat, pred = self._finish_step(i, t, eps, lat, text_cfg, img_cfg, guidance_rescale, stats, ref, noise_correct_step, flows)
            all_latent.append(lat[None])
            all_pred.append(pred[None])
            yield i
        return {"latent": lat[None], "all_latent": all_latent, "all_pred": all_pred}

    def _finish_step(self, i, t, eps, lat, text_cfg, img_cfg, guidance_rescale, stats, ref, noise_correct_step, flows, noise=None, bstride=0):
        """Everything of one sampling step behind the UNet for ONE clip: CFG combine (+ rescale), noise correction, scheduler
        step (inference.py:197-213, 270-277, 367-386).  eps: the clip's three branch predictions [3*F*h*w, 4] fp32, or (bstride > 0,
        the branch-major stack) a view that starts at its first branch with bstride fp32 elements between the branches."""
        dev = lat.device
        F, _, h, w = lat.shape
        if stats is not None:
            ops.cfg_stats(eps, stats, F, h, w, text_cfg, img_cfg, branch_stride=bstride)
        co = self.scheduler.coefficients(t)
        if self.scheduler.stochastic and co["coef"][3] != 0.0:
            if noise is None:
                inj = self.variance_noises[i] if self.variance_noises is not None else None
                noise = (inj[0].to(device=dev, dtype=torch.float32).contiguous() if inj is not None
                         else torch.randn(lat.shape, device=dev, dtype=torch.float32))
        else:
            noise = None
        new_lat, pred = torch.empty_like(lat), torch.empty_like(lat)
        correct = ref is not None and noise_correct_step * self.num_ddim_steps > i
        common = dict(text_cfg=text_cfg, img_cfg=img_cfg, sqrt_a=co["sqrt_a"], sqrt_1ma=co["sqrt_1ma"],
                      rescale_stats=stats, guidance_rescale=guidance_rescale, branch_stride=bstride)
        if correct and flows is not None:
            # combine -> flow-warped correction of the query frames -> step (inference.py:367-386)
            eps_cfg = torch.empty_like(lat)
            ops.cfg_step(eps, lat, nbranch=3, eps_out=eps_cfg, **common)
            dq = ops.flow_correction(eps_cfg, lat, ref, flows, co["sqrt_a"], co["sqrt_1ma"])
            ops.cfg_step(eps_cfg, lat, nbranch=0, coef=co["coef"], latent_out=new_lat, pred_x0=pred, latent_ref=ref,
                         correct=2, delta_q=dq, noise=noise, sqrt_a=co["sqrt_a"], sqrt_1ma=co["sqrt_1ma"])
        else:
            ops.cfg_step(eps, lat, nbranch=3, coef=co["coef"], latent_out=new_lat, pred_x0=pred,
                         latent_ref=ref if correct else None, correct=1 if correct else 0, noise=noise, **common)
        return new_lat, pred

    @torch.no_grad()
    def run_stacked(self, calls):
        """Run several independent ``__call__`` / ``second_clip_forward`` invocations (list of kwargs dicts as for
        ``run_concurrent``) as ONE batch: the 3 CFG branches of all n clips are stacked into every UNet launch
        (B = 3n; statistics stay per sample), so weights are read once per group of clips, every launch fills the chip
        and the lowest UNet levels need no split-K.  All clips must share shapes, ``start_time`` and the scheduler;
        guidance scales may differ.  Also the path of a batched ``__call__`` (inference.py:183-187 works for any b).
        (Round 4 measured two stacks of 5 clips running concurrently on two HIP streams against one stack of 10: 14.02 vs 14.50 frames/s -
        the persistent kernels own every CU, a second chain only fills tails and pays for it with half-sized launches.)"""
        n = len(calls)
        if n == 0:
            return []
        gen = self._stacked_gen(calls, 0)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                return done.value

    def _stacked_gen(self, calls, slot):
        """Generator behind run_stacked: yields after every DDIM step (so several stacks can be interleaved), returns the result dicts."""
        n = len(calls)
        # no more clips per launch chain than the kernels' 2 GiB operand window allows: larger stacks run as several,
        # as even as possible (every distinct stack size captures its own graph)
        Fc, hc, wc = calls[0]["latent"].shape[1], calls[0]["latent"].shape[-2], calls[0]["latent"].shape[-1]
        cap = max_clips_in_flight(Fc, hc, wc)
        if n > cap:
            ng = -(-n // cap)
            out, k = [], 0
            for g in range(ng):
                m = n // ng + (1 if g < n % ng else 0)
                out += yield from self._stacked_gen(calls[k:k + m], slot)
                k += m
            return out
        dev = self.unet.device
        st0 = calls[0].get("start_time", 0)
        clips = []
        for kw in calls:
            if kw.get("start_time", 0) != st0:
                raise ValueError("run_stacked: all clips must share start_time")
            lat, cond = self._clip_tensors(kw["latent"], kw["img_cond"])
            if clips and lat.shape != clips[0]["lat"].shape:
                raise ValueError("run_stacked: all clips must have the same [F,4,h,w]")
            ref = kw.get("latent_ref")
            gr = kw.get("guidance_rescale", 0.0)
            # optical-flow correction is per-clip elementwise work behind the shared UNet launch (inference.py:367-386): a clip's flows
            # ride with it (precomputed ``flows``, or ``ref_images`` / ``query_images`` for the pipe's estimator)
            fl = kw.get("flows")
            if fl is None and kw.get("ref_images") is not None:
                if kw["ref_images"].shape[0] != 1:
                    raise ValueError("only support batch size 1")   # inference.py:334
                if not hasattr(self, "obtain_flow_batched"):
                    raise ValueError("ref_images / query_images need the optical-flow pipe (InferenceIP2PVideoOpticalFlow)")
                fl = self.obtain_flow_batched(kw["ref_images"][0], kw["query_images"][0])
            if fl is not None:
                if ref is None:
                    raise ValueError("run_stacked: flows need latent_ref (second_clip_forward)")
                fl = self._latent_flows(fl, lat.shape[0] - ref.shape[1], lat.shape[-2], lat.shape[-1])
            clips.append(dict(lat=lat, cond=cond, flows=fl, ref=None if ref is None else ref[0].to(device=dev, dtype=torch.float32).contiguous(),
                              ncs=kw.get("noise_correct_step", 1.0) if ref is not None else 0.0,
                              text_cfg=kw.get("text_cfg", 7.5), img_cfg=kw.get("img_cfg", 1.2), gr=gr,
                              stats=torch.empty(2, device=dev, dtype=torch.float32) if gr > 0 else None,
                              noises=kw.get("noises"), all_latent=[], all_pred=[]))
        F, _, h, w = clips[0]["lat"].shape
        # BRANCH-major stack: sample br * n + c = branch br of clip c - the branches (no text, video) and (text, video), whose UNet inputs
        # are identical (inference.py:183-194), are the contiguous samples [n, 3n): the UNet computes their common prefix once (cfg_clips)
        ctx = torch.cat([torch.cat([kw["text_uncond"] for kw in calls], dim=0)] * 2 + [torch.cat([kw["text_cond"] for kw in calls], dim=0)], dim=0)
        runner = shared_runner(self.unet, 3 * n, F, h, w, ctx.shape[1], slot, self.use_graph, False, cfg_clips=n)
        runner.set_context(ctx)
        rows1 = F * h * w
        for i, t in enumerate(self.scheduler.timesteps[st0:]):
            t = int(t)
            for c, cl in enumerate(clips):
                ops.build_unet_input(cl["lat"], cl["cond"], runner.x_in[c * rows1:], runner.t[c:], t, 3, branch_rows=n * rows1, t_stride=n)
            eps = runner.run()
            for c, cl in enumerate(clips):
                noise = cl["noises"][i] if cl["noises"] is not None else None
                cl["lat"], pred = self._finish_step(i, t, eps[c * rows1:], cl["lat"], cl["text_cfg"], cl["img_cfg"], cl["gr"],
                                                    cl["stats"], cl["ref"], cl["ncs"], cl["flows"], noise=noise, bstride=n * rows1 * 4)
                cl["all_latent"].append(cl["lat"][None])
                cl["all_pred"].append(pred[None])
            yield
        return [{"latent": cl["lat"][None], "all_latent": cl["all_latent"], "all_pred": cl["all_pred"]} for cl in clips]

    def _batched_call(self, latent, text_cond, text_uncond, img_cond, latent_ref=None, **kw):
        """b > 1: the reference stacks the batch into the UNet call (inference.py:183-194); here every batch entry is a clip of
        ``run_stacked``.  A stochastic scheduler draws ONE [b,F,4,h,w] normal per step like the reference and slices it."""
        b = latent.shape[0]
        noises = None
        if self.scheduler.stochastic and self.variance_noises is None:
            dev = self.unet.device
            steps = len(self.scheduler.timesteps[kw.get("start_time", 0):])
            draws = [torch.randn(latent.shape, device=dev, dtype=torch.float32) for _ in range(steps)]
            noises = [[d[j].contiguous() for d in draws] for j in range(b)]
        elif self.variance_noises is not None:
            noises = [[(v[j] if v is not None else None) for v in self.variance_noises] for j in range(b)]
            noises = [[(x.to(device=self.unet.device, dtype=torch.float32).contiguous() if x is not None else None) for x in nj] for nj in noises]
        calls = []
        for j in range(b):
            c = dict(kw, latent=latent[j:j + 1], text_cond=text_cond[j:j + 1], text_uncond=text_uncond[j:j + 1], img_cond=img_cond[j:j + 1])
            if latent_ref is not None:
                c["latent_ref"] = latent_ref[j:j + 1]
            for k in ("ref_images", "query_images"):   # (the optical-flow pipe's batched form: one estimator pass per batch entry)
                if kw.get(k) is not None:
                    c[k] = kw[k][j:j + 1]
            if noises is not None:
                c["noises"] = noises[j]
            calls.append(c)
        res = self.run_stacked(calls)
        steps = len(res[0]["all_latent"])
        return {"latent": torch.cat([r["latent"] for r in res], 0),
                "all_latent": [torch.cat([r["all_latent"][k] for r in res], 0) for k in range(steps)],
                "all_pred": [torch.cat([r["all_pred"][k] for r in res], 0) for k in range(steps)]}

    @torch.no_grad()
    def run_concurrent(self, calls):
        """Run several independent ``__call__`` / ``second_clip_forward`` invocations (list of kwargs dicts, the
        optional key ``latent_ref`` selects ``second_clip_forward``) CONCURRENTLY: each gets its own HIP stream
        set and captured UNet graph and the steps are interleaved, so the GPU overlaps the kernels of
        different clips (units are independent: insv2v_run_loveu_tgve.py:83,101).  Returns the result dicts."""
        if not hasattr(self, "_slot_streams"):
            self._slot_streams = {}
        main = torch.cuda.current_stream()
        gens = []
        for slot, kw in enumerate(calls):
            kw = dict(kw)
            st = self._slot_streams.setdefault(slot, torch.cuda.Stream())
            st.wait_stream(main)
            args = dict(latent=kw["latent"], text_cond=kw["text_cond"], text_uncond=kw["text_uncond"], img_cond=kw["img_cond"],
                        text_cfg=kw.get("text_cfg", 7.5), img_cfg=kw.get("img_cfg", 1.2), start_time=kw.get("start_time", 0),
                        guidance_rescale=kw.get("guidance_rescale", 0.0), slot=slot)
            if kw.get("latent_ref") is not None:
                args.update(latent_ref=kw["latent_ref"], noise_correct_step=kw.get("noise_correct_step", 1.0))
            gens.append((st, self._loop_gen(**args)))
        results = [None] * len(gens)
        active = list(range(len(gens)))
        while active:
            for i in list(active):
                st, g = gens[i]
                with torch.cuda.stream(st):
                    try:
                        next(g)
                    except StopIteration as done:
                        results[i] = done.value
                        active.remove(i)
        for st, _ in gens:
            main.wait_stream(st)
        return results

    # ---- reference call surface ----------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, latent, text_cond, text_uncond, img_cond, text_cfg=7.5, img_cfg=1.2, start_time=0,
                 guidance_rescale=0.0):
        if latent.shape[0] != 1:
            return self._batched_call(latent, text_cond, text_uncond, img_cond, text_cfg=text_cfg, img_cfg=img_cfg, start_time=start_time,
                                      guidance_rescale=guidance_rescale)
        return self._loop(latent, text_cond, text_uncond, img_cond, text_cfg, img_cfg, start_time, guidance_rescale)

    @torch.no_grad()
    def second_clip_forward(self, latent, text_cond, text_uncond, img_cond, latent_ref, noise_correct_step=1.0,
                            text_cfg=7.5, img_cfg=1.2, start_time=0, guidance_rescale=0.0):
        if latent.shape[0] != 1:
            return self._batched_call(latent, text_cond, text_uncond, img_cond, latent_ref=latent_ref, noise_correct_step=noise_correct_step,
                                      text_cfg=text_cfg, img_cfg=img_cfg, start_time=start_time, guidance_rescale=guidance_rescale)
        return self._loop(latent, text_cond, text_uncond, img_cond, text_cfg, img_cfg, start_time, guidance_rescale,
                          latent_ref=latent_ref, noise_correct_step=noise_correct_step)


class InferenceIP2PVideoOpticalFlow(InferenceIP2PVideo):
    """Motion-compensated noise correction (inference.py:291-398).  The reference builds a torchvision RAFT estimator with downloaded
    weights here (inference.py:294, flow_utils.py:134-189).  This build carries the network itself on the HIP kernels (insv2v/raft.py,
    round 5) but no weights: pass ``raft_state_dict=`` (torchvision's ``raft_large`` checkpoint, its own key names) and the pipe builds
    ``RAFTFlow`` like the reference; or inject any ``flow_estimator(query[R,3,H,W], refs[R,3,H,W]) -> flow[R,2,H,W]`` (RAFTFlow's call
    signature); or hand precomputed ``flows=`` to ``second_clip_forward``."""

    def __init__(self, *args, flow_estimator=None, raft_state_dict=None, **kwargs):
        super().__init__(*args, **kwargs)
        if flow_estimator is None and raft_state_dict is not None:
            from .raft import RAFTFlow
            flow_estimator = RAFTFlow(self.unet.device).load_state_dict(raft_state_dict)
        self.flow_estimator = flow_estimator

    def obtain_flow_batched(self, ref_images, query_images):
        """One [R,2,H,W] flow set per query frame (inference.py:303-311: the query frame repeated against the R reference frames).  An
        estimator that advertises ``max_pairs`` (this build's RAFTFlow) gets several query frames per call: every (query, reference)
        pair is independent - InstanceNorm is per image, BatchNorm in eval mode - so the values are those of the per-query loop."""
        est = self.flow_estimator
        if est is None:
            raise RuntimeError("InferenceIP2PVideoOpticalFlow needs raft_state_dict= (the estimator's weights are not bundled), flow_estimator= or flows=")
        R = len(ref_images)
        per = getattr(est, "max_pairs", 0) // max(R, 1)
        if per >= 2:
            flows = []
            for i in range(0, len(query_images), per):
                qs = query_images[i:i + per]
                out = est(qs.repeat_interleave(R, dim=0), ref_images.repeat(len(qs), 1, 1, 1))
                flows += list(out.reshape(len(qs), R, *out.shape[1:]))
            return flows
        flows = []
        for q in query_images:
            flows.append(est(q.unsqueeze(0).repeat(R, 1, 1, 1), ref_images))
        return flows

    @torch.no_grad()
    def second_clip_forward(self, latent, text_cond, text_uncond, img_cond, latent_ref, ref_images=None,
                            query_images=None, noise_correct_step=1.0, text_cfg=7.5, img_cfg=1.2, start_time=0,
                            guidance_rescale=0.0, flows=None):
        if flows is None:
            assert ref_images.shape[0] == 1, "only support batch size 1"
            flows = self.obtain_flow_batched(ref_images[0], query_images[0])
        h, w = latent.shape[-2:]
        small = self._latent_flows(flows, latent.shape[1] - latent_ref.shape[1], h, w)
        return self._loop(latent, text_cond, text_uncond, img_cond, text_cfg, img_cfg, start_time, guidance_rescale,
                          latent_ref=latent_ref, noise_correct_step=noise_correct_step, flows=small)
