"""Data-parallel decoder training step around the raster hot path (BASELINE configs 3 and 5).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).  The raster
of one frame does not shard; the LATENT BATCH does: each rank decodes and renders `scenes_per_rank` scenes
(feature planes, positions, camera, target), then the flat fp32 gradient all-reduce (sum, / world) -- the reference's
own pattern eg3d/training/training_loop.py:288-299 -- and an Adam step.  Initial parameters are broadcast from rank
0 (training_loop.py:196).

The step mirrors main/train_pano2gaussian_decoder.py:217-265 with the parts that are out of scope replaced by
synthetic stand-ins of the same SIZE (SURVEY.md section 7 "hard parts", last item):
  * the GAN that produces feature planes and target images: a shared learnable plane tensor modulated per scene + a fixed
    synthetic target;
  * the finetuned generator backbone whose gradients the reference all-reduces (sequential_decoder_reverse.py:89-99,
    ~29.6 M parameters ~ 119 MB fp32): `backbone_params` floats that receive a dense gradient every step, are
    all-reduced and Adam-stepped with everything else;
  * the LPIPS term (main/loss_utils/lpips.py:6-34): losses.PerceptualStandIn, a fixed random VGG16-shaped trunk.
What is kept exactly: decoder -> GaussianModel attribute assignment (:223-227) -> CustomCam (:231) -> render_simple
(:232) -> L1 / L2 / SSIM / Sobel with the reference's weights (:36-40, :246-261) -> backward -> Adam (lr 9e-5, :32,213).

Host-side design for N GPUs: the gradients of all parameters live in ONE persistent flat fp32 buffer (p.grad are views),
cut into communication units of <= 32 MB; a unit's all-reduce is launched asynchronously FROM INSIDE the backward, by a
post-accumulate-grad hook, as soon as its last parameter's gradient is final (93 of the 119 MB -- the backbone's gradient
-- at the very start of the backward), and the Adam step of bucket k runs as soon as ITS units have finished.  Optionally
(scene_streams=True) the local scenes run on their own HIP streams, each with its own rasterizer context (ggd_ctx is
per (device, stream)); measured, it brings nothing on top of the single-call forward (see __init__) and is off.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .cameras import CustomCam, look_at_cam2world
from .decoder import SequentialDecoderReverse
from .losses import fused_image_loss, PerceptualStandIn
from .gaussian_model import GaussianModel

BUCKET_BYTES = 32 << 20


@dataclass
class SceneBatch:
    positions: torch.Tensor   # [B, N, 3]  surface points (reference: 500 000 per scene, target_dataloader.py:107)
    cam2world: torch.Tensor   # [B, 4, 4]
    fov_deg: torch.Tensor     # [B]
    target: torch.Tensor      # [B, 3, S, S]
    scene_id: torch.Tensor    # [B] global scene indices (which latent / plane set)


def make_scene_batch(scene_ids, n_points: int, size: int, device, seed: int = 0) -> SceneBatch:
    """Deterministic synthetic stand-in for TargetDataloader.get_data (target_dataloader.py:59-132): head-like
    shell positions, camera h,v ~ U around pi/2 (main/decoder_utils/camera.py:6-35), fov ~ U[5,17]
    (target_dataloader.py:71).  Each global scene id gets its own seed, so any rank can build any scene."""
    pos, c2w, fov, tgt = [], [], [], []
    for sid in scene_ids:
        g = torch.Generator().manual_seed(seed * 1000003 + int(sid))
        d = torch.randn(n_points, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        r = 0.3 * torch.clip(1.0 + 0.1 * torch.randn(n_points, 1, generator=g), 0.0, 1.0)
        pos.append(d * r)
        h = math.pi / 2 + (torch.rand(1, generator=g).item() * 2 - 1) * 1.0
        v = math.pi / 2 + (torch.rand(1, generator=g).item() * 2 - 1) * 0.3
        c2w.append(look_at_cam2world(h, v, 2.7))
        fov.append(5.0 + 12.0 * torch.rand(1, generator=g).item())
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
        blob = torch.exp(-(xx ** 2 + yy ** 2) * 3.0)
        col = torch.rand(3, generator=g)
        tgt.append(0.5 + (col[:, None, None] - 0.5) * blob[None])
    return SceneBatch(torch.stack(pos).to(device), torch.stack(c2w), torch.tensor(fov),
                      torch.stack(tgt).to(device), torch.tensor(list(scene_ids)))


class DecoderTrainer:
    """Holds the replicated decoder + shared feature planes (+ backbone stand-in), runs fwd/bwd for the local scenes, the
    bucketed gradient all-reduce and Adam."""

    def __init__(self, device, n_scenes_total: int, plane_res: int = 256, plane_channels: int = 32,
                 hidden_dim: int = 128, lr: float = 9e-5, image_size: int = 512, render_fn=None, seed: int = 0,
                 l1_weight: float = 0.2, l2_weight: float = 0.1, ssim_weight: float = 0.5, sobel_weight: float = 0.2,
                 loss_fn=None, process_group=None, fused_activations: bool = False, fused_decoder: bool = False,
                 backbone_params: int = 0, perceptual_weight: float = 0.0, perceptual_width_div: int = 1,
                 scene_streams: bool = False, decoder_precision: str = "bf16", plane_axes: str = "eg3d",
                 triplane_depth=None, fused_planes=None, force_comm=None):
        """plane_axes / triplane_depth: the generator whose planes are decoded -- ("eg3d", None): tri-planes
        [3, C, res, res]; ("panohead", 3): PanoHead's tri-grids [3, C * 3, res, res] sampled with a 3-D grid_sample
        (the reference's default generator: main/train_pano2gaussian_decoder.py:43, PanoHead/train.py:230,318,
        sequential_decoder_reverse.py:41-50).  fused_planes (default: on for CUDA): the scenes' planes
        `planes * latent` are never materialised -- one channel-last copy of the shared planes per step, the per-scene
        modulation applied inside the gather / scatter kernels, all scenes' plane gradients added into one buffer; channel
        counts the gather kernels do not take (not a power of two <= 64) keep the materialised-planes path.
        force_comm (default: the environment variable GGD_FORCE_COMM): take the collective path -- hooks, async all-reduce
        units, waits -- whenever a process group exists, even with ONE rank (exercises the RCCL communicator, work handles
        and their stream semantics on a single GPU; the sum over one rank is the identity)."""
        import os
        import torch.distributed as dist
        self.force_comm = bool(int(os.environ.get("GGD_FORCE_COMM", "0"))) if force_comm is None else bool(force_comm)
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.pg = process_group
        self.device = torch.device(device)
        self.image_size = image_size
        # loss weights: the reference's defaults (train_pano2gaussian_decoder.py:36-40)
        self.loss_w = dict(l1_weight=l1_weight, l2_weight=l2_weight, ssim_weight=ssim_weight, sobel_weight=sobel_weight)
        # loss_fn(image, target, **weights) -> (total, terms); default: the fused HIP kernel (csrc/ggd_imgloss.hip).  The
        # CPU (gloo) tests inject the torch evaluation from tests/.
        self.loss_fn = loss_fn if loss_fn is not None else fused_image_loss
        torch.manual_seed(seed)
        self.plane_axes, self.triplane_depth = plane_axes, (None if triplane_depth is None else int(triplane_depth))
        depth = self.triplane_depth or 1
        self.decoder = SequentialDecoderReverse(plane_channels, hidden_dim, plane_axes=plane_axes,
                                                triplane_depth=self.triplane_depth).to(self.device)
        # stand-in for the finetuned GAN backbone (shared, replicated, all-reduced like the reference's G): ONE learnable
        # tri-plane, modulated per scene by a fixed per-scene channel code (the "latent" z of that scene)
        g = torch.Generator().manual_seed(seed + 17)
        self.planes = torch.nn.Parameter(
            (0.5 * torch.randn(3, plane_channels * depth, plane_res, plane_res, generator=g)).to(self.device))
        self.latents = (1.0 + 0.25 * torch.randn(n_scenes_total, plane_channels * depth, generator=g)).to(self.device)
        self.plane_channels = plane_channels
        gather_ok = self.device.type == "cuda" and plane_channels <= 64 and (plane_channels & (plane_channels - 1)) == 0
        if fused_planes and not gather_ok:
            raise ValueError("fused_planes=True needs a HIP device and a power-of-two channel count <= 64")
        self.fused_planes = gather_ok if fused_planes is None else bool(fused_planes)
        # the rest of the backbone's gradient payload (see the module docstring): a dense gradient every step through a
        # fixed probe vector, so that its all-reduce and Adam step do real work
        self.backbone = None
        if backbone_params > 0:
            self.backbone = torch.nn.Parameter(torch.zeros(int(backbone_params), device=self.device))
            self.backbone_probe = torch.randn(int(backbone_params), generator=g).to(self.device)
        self.perceptual_weight = float(perceptual_weight)
        self.perceptual = None
        if self.perceptual_weight > 0:
            self.perceptual = PerceptualStandIn(seed=seed + 99, width_div=perceptual_width_div).to(self.device)
            if self.device.type == "cuda":
                self.perceptual = self.perceptual.to(memory_format=torch.channels_last)
        # fused_decoder: the MFMA decoder kernels (forward, activation backward, weight gradients) instead of the PyTorch
        # module; decoder_precision "bf16" (operands rounded to bf16) or "fp32" (split operands: the reference's precision)
        self.decoder_fwd = self.decoder
        self.fused_decoder = bool(fused_decoder)
        if fused_decoder:
            from .fused_decoder import FusedTrainDecoder
            self.decoder_fwd = FusedTrainDecoder(self.decoder, precision=decoder_precision)
        self.params = self.decoder.get_params_custom() + [self.planes] + ([self.backbone] if self.backbone is not None else [])
        self.broadcast_parameters()
        self._setup_flat_gradients(lr)
        self._setup_comm_units()
        if render_fn is None:
            from .gaussian_renderer import render_simple
            render_fn = render_simple
        self.render_fn = render_fn
        # fused_activations: sigmoid / exp / normalize inside the raster kernels (HIP render_simple only)
        self.render_kwargs = {"fused_activations": True} if fused_activations else {}
        self.bg = torch.tensor([0.55717, 0.52256, 0.51045], dtype=torch.float32, device=self.device)
        # scene_streams: every local scene's raster + loss on its own HIP stream (own ggd_ctx).  Measured on one MI355X
        # (4 scenes x 500 k points, fused decoder): 21.13 ms / step with and without -- the single-call forward already
        # enqueues the whole frame before it waits for num_rendered, so the GPU never idles on that read-back -- hence off
        # by default.  Only with the fused decoder (all scenes decoded in ONE launch on the main stream): with the PyTorch
        # decoder inside the per-scene streams autograd's cross-stream backward deadlocked on the second step.
        if scene_streams and not fused_decoder:
            raise ValueError("scene_streams=True needs fused_decoder=True")
        self.use_streams = bool(scene_streams) and self.device.type == "cuda"
        self._streams = []
        self.last_allreduce_bytes = 0
        self.last_allreduce_bytes_in_backward = 0
        self.measure_comm = False          # True: allreduce_and_step() brackets every unit's wait (allreduce_exposed_ms)

    # ---- gradients: one persistent flat buffer, bucketed --------------------------------------------------------------
    def _setup_flat_gradients(self, lr):
        """p.grad of every parameter is a view into self.flat_grad (autograd accumulates into an existing .grad in
        place); buckets are contiguous slices of <= BUCKET_BYTES (a parameter larger than that is split by the slice
        boundaries only for the all-reduce, it stays one Adam tensor of the bucket where it starts)."""
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        self.buckets = []          # (start, end, [params])
        cur_start, cur_params = 0, []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat_grad[off:off + n].view_as(p)
            cur_params.append(p)
            off += n
            if (off - cur_start) * 4 >= BUCKET_BYTES:
                self.buckets.append((cur_start, off, cur_params))
                cur_start, cur_params = off, []
        if cur_params:
            self.buckets.append((cur_start, off, cur_params))
        # an all-reduce chunk never exceeds BUCKET_BYTES: slice large buckets (one big parameter) for the collective only
        # one fused Adam kernel per bucket on the GPU (the default foreach form makes ~10 passes over the 119 MB of parameters)
        fused = torch.device(self.device).type == "cuda"
        self.optims = [torch.optim.Adam([{"params": ps, "lr": lr}], fused=fused) for _, _, ps in self.buckets]

    @property
    def world(self):
        return self.dist.get_world_size(self.pg) if self.dist else 1

    @property
    def _comm(self):
        """True when gradients travel through the collective path (more than one rank, or force_comm with a process group)."""
        return bool(self.dist and (self.world > 1 or self.force_comm))

    @property
    def rank(self):
        return self.dist.get_rank(self.pg) if self.dist else 0

    def broadcast_parameters(self):
        if self.dist:
            for p in self.params:
                self.dist.broadcast(p.data, src=0, group=self.pg)

    # ---- gradient all-reduce, overlapped with the backward ---------------------------------------------------------------
    def _setup_comm_units(self):
        """Cut the flat gradient into communication units (contiguous slices of <= BUCKET_BYTES) and register a
        post-accumulate-grad hook on every parameter: a unit's all-reduce is launched -- asynchronously, from inside the
        backward -- as soon as the LAST parameter overlapping it has its final gradient, so the collective runs under the
        rest of the backward instead of after it (the reference all-reduces after the backward,
        eg3d/training/training_loop.py:288-299; same result, the sum is only started earlier).  The bulk of the payload,
        the backbone's gradient (93 of the 119 MB), is final at the very start of the backward; what remains at its end is
        the decoder's + the planes' unit (26 MB).  Only set up when a process group with more than one rank exists."""
        chunk = max(1, BUCKET_BYTES // 4)
        self.units = []                       # [start, end, bucket index, params still missing this step, work]
        self._units_of_param = {}
        for k, (s, e, _) in enumerate(self.buckets):
            for c0 in range(s, e, chunk):
                self.units.append(dict(start=c0, end=min(e, c0 + chunk), bucket=k, need=0, missing=0, work=None))
        off = 0
        for p in self.params:
            n = p.numel()
            mine = [u for u in self.units if u["start"] < off + n and u["end"] > off]
            for u in mine:
                u["need"] += 1
            self._units_of_param[id(p)] = mine
            off += n
        self._hooks = []
        if self._comm:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._grad_ready))
        self._arm_units()

    def _arm_units(self):
        """Ready for the next backward.  Collectives a failed step left in flight are waited for first (every rank
        launched them in the same order, so they complete), never dropped: a stale handle would otherwise be waited on --
        and a new collective skipped -- by the next step, and the ranks would diverge silently."""
        for u in self.units:
            if u["work"] is not None:
                # (bounded: if only THIS rank's step failed, the peers never launched the matching collectives, and an unbounded
                # wait would sit here until the backend's own timeout instead of surfacing the error that caused it)
                import datetime
                try:
                    ok = u["work"].wait(timeout=datetime.timedelta(seconds=60))
                except TypeError:                      # (a backend whose Work.wait takes no timeout)
                    ok = u["work"].wait()
                if ok is False:
                    raise RuntimeError("DecoderTrainer: an all-reduce left in flight by a failed step did not complete within "
                                       "60 s -- the ranks have diverged (did only this rank's step fail?)")
            u["missing"], u["work"] = u["need"], None
        self._launched_bytes = 0
        self._launched_in_backward = 0
        self._in_step_tail = False

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it add into the flat gradient without
        launching any all-reduce; the backward that completes the accumulation runs outside it (its hooks launch the
        units as usual), or allreduce_and_step() launches whatever is left."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._defer = getattr(self, "_defer", False), True
            try:
                yield
            finally:
                self._defer = prev
                for u in self.units:           # the next backward counts every parameter again
                    if u["work"] is None:
                        u["missing"] = u["need"]
        return cm()

    def _launch_unit(self, u):
        u["work"] = self.dist.all_reduce(self.flat_grad[u["start"]:u["end"]], op=self.dist.ReduceOp.SUM, group=self.pg,
                                         async_op=True)
        self._launched_bytes += (u["end"] - u["start"]) * 4
        self._launch_seq = getattr(self, "_launch_seq", 0) + 1
        u["seq"] = self._launch_seq

    def _grad_ready(self, p):
        if getattr(self, "_defer", False):
            return
        for u in self._units_of_param[id(p)]:
            if u["work"] is not None or u["missing"] <= 0:
                # a second backward would add into a slice whose all-reduce is already in flight (or done) and never be reduced
                raise RuntimeError("DecoderTrainer: exactly one backward per allreduce_and_step() -- this parameter's "
                                   "gradient was already handed to the all-reduce; accumulate under trainer.no_sync()")
            u["missing"] -= 1
            if u["missing"] == 0:
                self._launch_unit(u)
                self._launched_in_backward += (u["end"] - u["start"]) * 4

    def allreduce_and_step(self):
        """Per bucket: wait for its units' all-reduces (launched from the backward by the hooks; any unit a hook did not
        reach -- a parameter without a gradient this step -- is launched here) -> / world -> sanitise -> Adam.  Bucket k's Adam
        runs while the all-reduces of the later buckets are still in flight."""
        world = self.world
        multi = self._comm
        if multi:
            for u in self.units:
                if u["work"] is None:
                    self._launch_unit(u)
        stalls = []
        # buckets in the order their LAST unit was launched (the same on every rank: hooks fire in the autograd graph's order):
        # the backbone's gradient was final -- and its all-reduce launched -- at the start of the backward, the planes' and
        # the decoder's at its very end, so the backbone buckets' Adam runs while the late units are still travelling
        order = list(range(len(self.buckets)))
        if multi:
            last = {k: max(u.get("seq", 0) for u in self.units if u["bucket"] == k) for k in order}
            order.sort(key=lambda k: last[k])
        for k in order:
            s, e, _ = self.buckets[k]
            g = self.flat_grad[s:e]
            if multi:
                for u in self.units:
                    if u["bucket"] == k:
                        if self.measure_comm:
                            stalls.append(self._timed_wait(u["work"]))
                        else:
                            u["work"].wait()
                g /= world
            # the reference sanitises on every step, single-GPU runs included (eg3d/training/training_loop.py:288-299)
            torch.nan_to_num(g, nan=0.0, posinf=1e5, neginf=-1e5, out=g)
            self.optims[k].step()
        nbytes = self._launched_bytes if multi else 0
        if self.measure_comm:
            self._pending_stalls = stalls      # resolved lazily (allreduce_exposed_ms): reading an event time synchronises
        self.last_allreduce_bytes = nbytes
        self.last_allreduce_bytes_in_backward = self._launched_in_backward if multi else 0
        self._arm_units()
        return nbytes

    def _timed_wait(self, work):
        """work.wait() bracketed so that the time the COMPUTE stream stood still for this collective can be read later:
        a HIP event pair on the current stream (RCCL: wait() only makes the stream wait), wall clock on the CPU (gloo)."""
        if self.device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            return (e0, e1)
        import time
        t0 = time.perf_counter()
        work.wait()
        return (time.perf_counter() - t0) * 1e3

    @property
    def allreduce_exposed_ms(self):
        """With measure_comm = True: how long the last step's compute stream was stalled waiting for all-reduce units --
        what of the collective was NOT hidden under the backward / the earlier buckets' Adam (0 on one rank).  Reading it
        synchronises the device."""
        tot = 0.0
        for st in getattr(self, "_pending_stalls", []):
            if isinstance(st, tuple):
                st[1].synchronize()
                tot += st[0].elapsed_time(st[1])
            else:
                tot += st
        return tot

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _scene_features(self, batch, scene_ids):
        """Plane-mean features of all local scenes [B * N, C] from ONE channel-last copy of the shared planes, the scenes'
        latent modulation applied inside the gather (fused_planes)."""
        from .decoder import planes_channels_last, planes_gather
        planes_cl = planes_channels_last(self.planes, self.triplane_depth)
        depth = self.triplane_depth or 1
        mods = self.latents[scene_ids].view(len(scene_ids), self.plane_channels, depth).transpose(1, 2).contiguous()
        return planes_gather(planes_cl, batch.positions, self.decoder.box_warp, self.plane_axes, self.triplane_depth, mod=mods)

    def _scene_loss(self, batch, b, scene_id, attrs, feats=None):
        gs = GaussianModel(0)    # one container per in-flight scene (its tensors are saved by autograd until backward)
        if attrs is not None:
            gs._xyz, gs._scaling, gs._rotation, gs._opacity, color = attrs[b]
            gs._features_dc = color.unsqueeze(1)
        else:
            if feats is not None:
                out = self.decoder_fwd(None, batch.positions[b], features=feats[b])
            else:
                planes = self.planes * self.latents[scene_id][None, :, None, None]
                out = self.decoder_fwd(planes, batch.positions[b])
            gs._xyz, gs._scaling, gs._rotation = out.xyz, out.scale, out.rotation
            gs._opacity, gs._features_dc = out.opacity, out.color.unsqueeze(1)
        fov = float(batch.fov_deg[b]) / 360 * 2 * math.pi
        # the 4x4 algebra of the camera prologue on the host (a device-side inverse is a solver call with a host
        # sync and ~40 tiny launches per scene), the three matrices uploaded once
        cam = CustomCam(size=self.image_size, fov=fov, extr=batch.cam2world[b].cpu())
        for name in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(cam, name, getattr(cam, name).to(self.device, non_blocking=True))
        image = self.render_fn(cam, gs, bg_color=self.bg, **self.render_kwargs)["render"][:3]
        target = batch.target[b]
        loss = self.loss_fn(image, target, **self.loss_w)[0]
        if self.perceptual is not None:
            # the reference evaluates its perceptual network on the whole batch at once (lpips.py:29-31): the rendered
            # images are collected and local_loss makes ONE call on the stack
            self._perc_images.append(image)
        return loss

    def local_loss(self, batch: SceneBatch):
        """Decoder + raster forward for the local scenes; returns the mean loss over them."""
        B = batch.positions.shape[0]
        scene_ids = batch.scene_id.tolist()
        attrs = None
        feats = self._scene_features(batch, scene_ids) if self.fused_planes else None
        if feats is not None and not self.fused_decoder:
            # per-scene rows for the PyTorch decoder: unbind's backward is ONE stack (slices would each zero-fill and add a
            # full [B * N, C] gradient)
            feats = feats.view(B, batch.positions.shape[1], -1).unbind(0)
        if self.fused_decoder:   # all local scenes through one decoder launch
            from .fused_decoder import split_attrs
            planes_list = None if feats is not None else [self.planes * self.latents[s][None, :, None, None] for s in scene_ids]
            attrs = split_attrs(self.decoder_fwd.forward_scenes(planes_list, batch.positions, feats=feats))
        losses = []
        self._perc_images = []
        if self.use_streams and B > 1:
            main = torch.cuda.current_stream(self.device)
            while len(self._streams) < B:
                self._streams.append(torch.cuda.Stream(device=self.device))
            for b in range(B):
                st = self._streams[b]
                st.wait_stream(main)                    # the decoder outputs / parameters come from the main stream
                with torch.cuda.stream(st):
                    losses.append(self._scene_loss(batch, b, scene_ids[b], attrs, feats))
            for b in range(B):
                main.wait_stream(self._streams[b])
                losses[b].record_stream(main)
        else:
            for b in range(B):
                losses.append(self._scene_loss(batch, b, scene_ids[b], attrs, feats))
        total = losses[0]
        for l in losses[1:]:
            total = total + l
        if self.perceptual is not None:
            total = total + self.perceptual_weight * self.perceptual(torch.stack(self._perc_images), batch.target[:B])
            self._perc_images = []
        total = total / B
        if self.backbone is not None:
            total = total + 1e-8 * torch.dot(self.backbone, self.backbone_probe)
        return total

    def step(self, batch: SceneBatch) -> float:
        self._arm_units()          # a previous step that raised inside its backward left units half-counted / in flight
        self.flat_grad.zero_()
        loss = self.local_loss(batch)
        loss.backward()
        self.allreduce_and_step()
        return float(loss.detach())
