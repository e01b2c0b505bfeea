"""The benchmark's workloads, built from the deterministic synthetic generator (`moda_amd/synth.py`): the `models` / `embeddings`
dicts of BASELINE.json's configurations on the GPU, and `TrainHarness` -- one rank's training step at configs[3] size as the
reference's trainer runs it.  `bench.py` and the GPU tests (through tests/gpu_helpers.py) use the SAME objects, so what the tests
check is what the bench times.  Not part of the drop-in surface."""
import types

import numpy as np
import torch

import moda_amd
from . import synth

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def nerf_from_params(p, **kw):
    m = moda_amd.NeRF(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m.to(DEV).eval()


def make_models(seed, B, with_skin=True, with_feat=False, with_vis=False, alpha=10.0, perturb_bones=False, beta=0.1,
                with_dis=False):
    mp = synth.make_models(seed, B=B, with_skin=with_skin, with_feat=with_feat, with_vis=with_vis,
                           perturb_bones=perturb_bones, beta=beta, with_dis=with_dis)
    models = {"coarse": nerf_from_params(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=beta)}
    if B > 0:
        models["bones"] = torch.nn.Parameter(T(mp["bones_rst"]))
        models["bones_rst"] = T(mp["bones_rst"])
        models["skin_aux"] = T(mp["skin_aux"])
        rpc = torch.nn.Embedding(1, 128).to(DEV)
        if with_skin:
            models["nerf_skin"] = nerf_from_params(mp["nerf_skin"], D=5, W=64, in_channels_xyz=63 + 128,
                                                   in_channels_dir=0, out_channels=B, raw_feat=True,
                                                   in_channels_code=128)
            rpc.weight.data = T(mp["rest_pose_code"])
        models["rest_pose_code"] = rpc
    if with_dis:
        models["nerf_dis"] = nerf_from_params(mp["nerf_dis"], D=5, W=128, in_channels_xyz=63 + 128, in_channels_dir=0,
                                              out_channels=3, raw_feat=True, in_channels_code=128)
    if with_feat:
        models["nerf_feat"] = nerf_from_params(mp["nerf_feat"], D=5, W=128, in_channels_xyz=63, in_channels_dir=0,
                                               out_channels=16, raw_feat=True, init_beta=1.0)
    if with_vis:
        models["nerf_vis"] = nerf_from_params(mp["nerf_vis"], D=5, W=64, in_channels_xyz=63, in_channels_dir=0,
                                              out_channels=1, raw_feat=True)
    emb = {"xyz": moda_amd.Embedding(3, 10, alpha=alpha), "dir": moda_amd.Embedding(3, 4, alpha=alpha)}
    return models, emb


def make_opts(**kw):
    o = dict(dist_corresp=False, lbs=False, neudbs=True, symm_shape=False, scale_rgb=1.3, rgb_filter=False,
             use_corresp=False, use_corr=False, use_ot=False, s3im_loss=False)
    o.update(kw)
    return types.SimpleNamespace(**o)


def rays_to_gpu(rays):
    return {k: T(v) for k, v in rays.items()}


# ---- the reference's training step at BASELINE configs[3] size, shared by bench.py --mode train and the GPU tests ------------
TRAIN_TERMS = ("img", "sil", "frnd", "flo", "feat", "proj", "vis", "cyc")
TRAIN_WEIGHTS = dict(img_wt=1.0, sil_wt=0.1, frnd_wt=0.01, flow_wt=1.0, feat_wt=0.01, proj_wt=0.02, vis_wt=1.0, cyc_wt=0.05)
# what the no-grad coarse pre-pass of a hierarchical (use_fine) training step runs in, per training precision: the inference
# kernels of the same accuracy class (bf16 training: the bf16 throughput kernels; the fp32-grade modes: split-bf16 / exact fp32)
PREPASS_PRECISION = {"bf16": "bf16", "bf16x3": "bf16x3", "bf16x6": "bf16x3", "fp32": "fp32"}


class TrainHarness:
    """One rank's training step as the reference's trainer runs it (nnutils/train_utils.py:950-969): forward + backward of
    the total loss assembled as nnutils/moda.py:540-640 does (default weights), DDP-style gradient all-reduce for world > 1,
    AdamW(betas (0.9, 0.999), weight_decay 1e-4, train_utils.py:227-250).  The learning rate is what OneCycleLR(max_lr 5e-4,
    div_factor 25, train_utils.py:260-288) applies at the start of training: 5e-4 / 25 = 2e-5 (`lr`).

    Everything about a step is a deterministic function of (seed, step index): the rays are fixed, the two random tensors a
    step draws (depth jitter, visibility-loss negatives) come from this object's own generator, also under graph replay.
    `terms` (device, 8 floats) holds the weighted loss terms of the last step in TRAIN_TERMS order; `loss_buf` = [loss * N, N]."""

    def __init__(self, N=2048, S=128, B=25, precision="bf16", rank=0, world=1, dist=None, lr=2e-5, device=None, seed=1000,
                 rays_per_frame=4, fused_adamw=True, bucket=True, use_fine=False, with_unc=False, strong=False):
        """use_fine / with_unc: the reference's LAST training stage (scripts/template.sh:59: --fine_steps 0 --use_unc): S/2 coarse
        depths rendered without gradients (rendering.py:91-107, here on the fused inference kernels in `PREPASS_PRECISION[precision]`),
        S/2 importance samples merged in, and the uncertainty network nerf_unc (8x256, moda.py:457-464) trained on
        | sil * img_loss - unc_pred |^2 (moda.py:707-720) -- BASELINE configs[4] as a training step."""
        from moda_amd import sharding
        global DEV
        # strong=True: ONE batch of N rays (the one-rank run's rays) cut into contiguous per-rank ranges (sharding.shard_rays), so
        # that the union of the ranks' rays is the single-rank batch; default: every rank owns its own N rays (weak, DDP's lines)
        n_total = N
        if strong:
            lo, hi = sharding.shard_bounds(n_total, rank, world, align=rays_per_frame)
            N = hi - lo
        self.N, self.S, self.B, self.world, self.dist = N, S, B, world, dist
        self.dev = device or DEV
        self.precision = precision
        self.use_fine, self.with_unc = bool(use_fine), bool(with_unc)
        moda_amd.set_train_precision(precision)
        self.prepass_precision = PREPASS_PRECISION[precision]
        # MoDA's default training configuration (moda.py:60-173): coarse + skin + CSE feature + visibility nets, paired-frame
        # correspondence (dist_corresp), Sinkhorn feature matching (use_ot), all per-ray loss keys present
        prev, DEV = DEV, self.dev
        try:
            self.models, self.emb = make_models(0, B, with_feat=True, with_vis=True)
            sd = seed if strong else sharding.rank_seed(seed, rank)
            n_make = n_total if strong else N
            rays = dict(synth.make_rays(sd, n_make, B, rays_per_frame=rays_per_frame))
            rays.update(synth.make_corresp_rays(sd, n_make, B, rays_per_frame=rays_per_frame))
            rays.update(synth.make_feat_rays(sd, n_make, rays_per_frame=rays_per_frame))
            if self.with_unc:
                rays.update(synth.make_unc_rays(sd, n_make, rays_per_frame))
            if strong:
                rays = sharding.shard_rays({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in rays.items()}, rank, world)
                rays = {k: v.numpy() for k, v in rays.items()}
            rays = rays_to_gpu(rays)
            if self.with_unc:
                unc = moda_amd.NeRFUnc(in_channels_xyz=63, D=8, W=256, out_channels=1, in_channels_dir=32, raw_feat=True, init_beta=1.)
                unc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(
                    0, "nerf_unc", D=8, W=256, in_channels_xyz=63, in_channels_dir=32, out_channels=1, init_beta=1.).items()})
                self.models["nerf_unc"] = unc.to(self.dev)
        finally:
            DEV = prev
        for m in self.models.values():
            if isinstance(m, torch.nn.Module):
                m.train()
        self.models["bones_rst"] = torch.nn.Parameter(self.models["bones_rst"].clone())
        self.models["skin_aux"] = torch.nn.Parameter(self.models["skin_aux"].clone())
        for k in ("bone_rts", "bone_rts_target", "bone_rts_dentrg", "time_embedded", "env_code", "rays_o", "rays_d", "rtk_vec",
                  "rtk_vec_target", "rtk_vec_dentrg"):
            rays[k].requires_grad_(True)
        self.rays = rays
        self.params = [p for m in self.models.values() if isinstance(m, torch.nn.Module) for p in m.parameters()]
        self.params += [self.models["bones_rst"], self.models["skin_aux"]]
        self.opts = make_opts(dist_corresp=True, use_corresp=True, use_ot=True)
        self.bound = np.asarray([0.2, 0.2, 0.2], np.float32)
        self.loss_buf = torch.zeros(2, device=self.dev)
        self.terms = torch.zeros(len(TRAIN_TERMS), device=self.dev)
        kw = dict(lr=lr, betas=(0.9, 0.999), weight_decay=1e-4, capturable=True)
        self.opt = None
        if fused_adamw:     # one fused kernel per step (the foreach form issues ~150 one-element divisions for its bias corrections)
            try:
                self.opt = torch.optim.AdamW(self.params, fused=True, **kw)
            except (RuntimeError, TypeError, ValueError):
                self.opt = None
        if self.opt is None:
            self.opt = torch.optim.AdamW(self.params, **kw)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed * 7919 + rank)
        self.vis_neg = torch.empty((1, N * S, 3), device=self.dev)      # negatives of the visibility loss (loss_utils.py:137)
        S_c = S // 2 if self.use_fine else S                            # coarse depths of a hierarchical step (rendering.py:50)
        self.jitter = torch.empty((N, S_c), device=self.dev)            # depth jitter (rendering.py:82)
        self.pdf_u = torch.empty((N, S_c), device=self.dev) if self.use_fine else None    # resampling uniforms (:607)
        self.noise_pre = torch.zeros((N, S_c), device=self.dev) if self.use_fine else None
        self.feat_noise = torch.empty((1, 8000, 3), device=self.dev)    # lattice jitter of feat_match (loss_utils.py:306)
        self.noise_raw = torch.zeros((N, S), device=self.dev)           # density noise (rendering.py:193): noise_std is 0
        self.graph = self.graph_tail = None
        self.graph_form = "eager"
        self.steps_done = 0
        # bucket: after the first step has shown which network parameters receive a gradient (the heads a network does not
        # evaluate get none, and AdamW must keep skipping them), those parameters' gradients become views of ONE flat buffer
        # that the backward kernels add into directly (moda_amd.GradBucket)
        self.want_bucket = bucket
        self.bucket = None

    @staticmethod
    def _masked_mean(x, m):            # x[m].mean() without the boolean gather (no host sync, graph-capturable)
        m = m.to(x.dtype).expand_as(x)
        return (x * m).sum() / m.sum()

    def draw(self):
        """The step's random tensors, outside any graph (a captured generator needs registration; this does not)."""
        self.vis_neg.uniform_(generator=self.gen)
        self.jitter.uniform_(generator=self.gen)
        self.feat_noise.normal_(generator=self.gen)
        if self.pdf_u is not None:
            self.pdf_u.uniform_(generator=self.gen)

    def fwd_bwd(self):
        from moda_amd.loss_utils import total_loss, unc_loss
        from moda_amd.nerf import precision_scope
        with precision_scope(self.prepass_precision if self.use_fine else None):
            r = moda_amd.render_rays(self.models, self.emb, self.rays, N_samples=self.S, perturb=1.0, noise_std=0.0, opts=self.opts,
                                     img_size=512, obj_bound=self.bound, use_fine=self.use_fine,
                                     rng={"vis_neg_rand": self.vis_neg, "perturb_rand": self.jitter, "feat_noise": self.feat_noise,
                                          "noise_raw": self.noise_raw, "pdf_u": self.pdf_u, "noise_raw_pre": self.noise_pre})
        # moda.py:540-705 as one launch each way; these weights (not the flags' defaults) keep every term of the synthetic scene
        # within two orders of magnitude of the others, and the loss values comparable across rounds
        loss, terms = total_loss(r, TRAIN_WEIGHTS)
        if self.with_unc:
            loss = loss + unc_loss(r)                                  # moda.py:707-720
        loss.backward()
        self.terms.copy_(torch.stack([terms[k] for k in TRAIN_TERMS]))
        return loss.detach()

    def zero_grad(self):
        """Every gradient gone: the bucket by one memset (recorded into a captured graph like any launch), the rest set to None."""
        if self.bucket is None:
            self.opt.zero_grad(set_to_none=True)
            return
        inb = {id(p) for p in self.bucket.params}
        for p in self.params:
            if id(p) not in inb:
                p.grad = None
        self.bucket.attach()
        self.bucket.zero()

    def _make_bucket(self):
        """Every parameter that received a gradient in the first step -- the networks' (written directly by NerfFn's backward) and
        the small ones autograd accumulates (bones_rst, skin_aux, the rest-pose code) -- plus the two loss sums, in ONE flat
        buffer: a step's whole exchange is one all-reduce."""
        nets = [p for p in self.params if p.grad is not None]
        old = [p.grad for p in nets]
        self.bucket = moda_amd.GradBucket(nets, extra=2)
        for p, g in zip(nets, old):                  # this step's gradients move into their views
            p.grad.copy_(g)
        self.bucket.extra.copy_(self.loss_buf)
        self.loss_buf = self.bucket.extra

    def eager_step(self):
        from moda_amd import sharding
        self.draw()
        self.zero_grad()
        loss = self.fwd_bwd()
        if self.want_bucket and self.bucket is None:
            self._make_bucket()                      # this step's gradients have moved into the bucket's views: exchanged below
        self.loss_buf[0] = loss * self.N
        self.loss_buf[1] = float(self.N)
        if self.bucket is not None:
            self.bucket.all_reduce(self.dist, self.world, force=sharding.COLLECTIVES_AT_WORLD_1)   # gradients (mean) + loss sums: ONE ~11 MB collective
            # a parameter outside the bucket that receives a gradient after all (it had none in step 1): exchanged like the rest,
            # never applied un-averaged
            inb = self._in_bucket()
            late = [p for p in self.params if id(p) not in inb and p.grad is not None]
            if late:
                sharding.allreduce_gradients(late, self.dist, self.world)
        else:
            sharding.allreduce_gradients(self.params, self.dist, self.world)
            sharding.allreduce_sums(self.loss_buf, self.dist, self.world)
        self.opt.step()
        self.steps_done += 1
        return self.loss_buf

    def _in_bucket(self):
        return {id(p) for p in self.bucket.params} if self.bucket is not None else set()

    def capture(self, warm=3):
        """The step as HIP graphs, replayed -- it is launch-latency-bound when issued eagerly (~440 launches for ~6 ms of kernel
        time).  `warm` eager steps run first on a side stream (they count as steps).
        One rank: ONE graph (zero the bucket, forward, backward, AdamW).
        Several ranks: TWO graphs around the step's single collective -- [zero, forward, backward, loss sums] | all-reduce of the
        flat bucket (RCCL, eager: one call) | [mean, AdamW].  The collective stays outside the graphs: its capture works in
        thread-local capture mode only (the process group's watchdog thread polls events, which a GLOBAL-mode capture forbids
        process-wide; tools/rccl_probe.py), and MODA_GRAPH_COLLECTIVE=1 selects that single-graph form."""
        import os
        from moda_amd import sharding
        collective = sharding.live(self.world)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.bucket is not None:
            # the captured step exchanges the bucket and nothing else: every gradient the warm-up steps produced must live in it,
            # and every rank must hold the same layout
            inb = self._in_bucket()
            stray = [i for i, p in enumerate(self.params) if id(p) not in inb and p.grad is not None]
            if stray:
                raise RuntimeError(f"TrainHarness.capture: parameters {stray} receive gradients outside the gradient bucket")
            if collective:
                self.bucket.check_same_layout_on_all_ranks(self.dist, self.world)
        self.zero_grad()
        n_f = float(self.N)

        def body():
            if self.bucket is not None:
                self.bucket.zero()                    # the memset is part of the replayed step
            g_loss = self.fwd_bwd()
            self.loss_buf[0] = g_loss * self.N
            self.loss_buf[1:2].fill_(n_f)             # (a fill kernel with a constant: no host scalar inside the capture)

        if not collective:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
                self.opt.step()
            self.graph, self.graph_tail, self.graph_form = graph, None, "one graph"
            return graph
        if self.bucket is None:
            raise RuntimeError("TrainHarness.capture with several ranks needs the gradient bucket (bucket=True)")
        if os.environ.get("MODA_GRAPH_COLLECTIVE") == "1":
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                body()
                self.bucket.all_reduce(self.dist, self.world, force=True)
                self.opt.step()
            self.graph, self.graph_tail, self.graph_form = graph, None, "one graph with the all-reduce inside"
            return graph
        head, tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(head, capture_error_mode="thread_local"):
            body()
        with torch.cuda.graph(tail, capture_error_mode="thread_local"):
            self.bucket.scale(self.world)
            self.opt.step()
        self.graph, self.graph_tail, self.graph_form = head, tail, "two graphs around one eager all-reduce"
        # the captures themselves execute nothing: parameters and optimiser state are those after `warm` steps
        return head

    def step(self):
        if self.graph is None:
            return self.eager_step()
        self.draw()
        self.graph.replay()
        if self.graph_tail is not None:
            self.bucket.all_reduce(self.dist, self.world, average=False, force=True)
            self.graph_tail.replay()
        self.steps_done += 1
        return self.loss_buf

    def loss(self):
        return float(self.loss_buf[0] / self.loss_buf[1])

    def invalidate_inference_caches(self):
        for m in self.models.values():
            if hasattr(m, "invalidate_packed"):
                m.invalidate_packed()
