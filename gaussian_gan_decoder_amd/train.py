"""Data-parallel decoder training step around the raster hot path (BASELINE configs 3 and 5).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).  The raster
of one frame does not shard; the LATENT BATCH does: each rank decodes and renders `scenes_per_rank` scenes
(feature planes, positions, camera, target), then ONE flat fp32 gradient all-reduce (sum, / world) -- the reference's
own pattern eg3d/training/training_loop.py:288-299 -- and an Adam step.  Initial parameters are broadcast from rank
0 (training_loop.py:196).

The step mirrors main/train_pano2gaussian_decoder.py:217-265 with the parts that are out of scope replaced by
synthetic stand-ins (SURVEY.md section 7 "hard parts", last item): the GAN that produces feature planes and target
images is a per-scene learnable plane tensor + a fixed synthetic target; the loss is L1 (+ L2), the only image
losses without external networks.  What is kept exactly: decoder -> GaussianModel attribute assignment (:223-227)
-> CustomCam (:231) -> render_simple (:232) -> loss -> backward -> Adam (lr 9e-5, :32,213).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .cameras import CustomCam, look_at_cam2world
from .decoder import SequentialDecoderReverse
from .losses import fused_image_loss, image_loss_torch
from .gaussian_model import GaussianModel


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
    """Holds the replicated decoder + per-scene feature planes, runs fwd/bwd for the local scenes and the
    flat gradient all-reduce."""

    def __init__(self, device, n_scenes_total: int, plane_res: int = 256, plane_channels: int = 32,
                 hidden_dim: int = 128, lr: float = 9e-5, image_size: int = 512, render_fn=None, seed: int = 0,
                 l1_weight: float = 0.2, l2_weight: float = 0.1, ssim_weight: float = 0.5, sobel_weight: float = 0.2,
                 fused_loss: bool = False, process_group=None, fused_activations: bool = False,
                 fused_decoder: bool = False):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.pg = process_group
        self.device = torch.device(device)
        self.image_size = image_size
        # loss weights: the reference's defaults (train_pano2gaussian_decoder.py:36-40); the LPIPS / identity terms need
        # external networks and are out of scope
        self.loss_w = dict(l1_weight=l1_weight, l2_weight=l2_weight, ssim_weight=ssim_weight, sobel_weight=sobel_weight)
        self.fused_loss = bool(fused_loss)   # csrc/ggd_imgloss.hip instead of the torch conv graph
        torch.manual_seed(seed)
        self.decoder = SequentialDecoderReverse(plane_channels, hidden_dim).to(self.device)
        # stand-in for the finetuned GAN backbone (shared, replicated, all-reduced like the reference's G): ONE learnable
        # tri-plane, modulated per scene by a fixed per-scene channel code (the "latent" z of that scene)
        g = torch.Generator().manual_seed(seed + 17)
        self.planes = torch.nn.Parameter(
            (0.5 * torch.randn(3, plane_channels, plane_res, plane_res, generator=g)).to(self.device))
        self.latents = (1.0 + 0.25 * torch.randn(n_scenes_total, plane_channels, generator=g)).to(self.device)
        # fused_decoder: bf16-MFMA decoder kernels (forward + activation backward) instead of the PyTorch module
        self.decoder_fwd = self.decoder
        self.fused_decoder = bool(fused_decoder)
        if fused_decoder:
            from .fused_decoder import FusedTrainDecoder
            self.decoder_fwd = FusedTrainDecoder(self.decoder)
        self.params = self.decoder.get_params_custom() + [self.planes]
        self.broadcast_parameters()
        self.optim = torch.optim.Adam([{"params": self.params, "lr": lr}])
        if render_fn is None:
            from .gaussian_renderer import render_simple
            render_fn = render_simple
        self.render_fn = render_fn
        # fused_activations: sigmoid / exp / normalize inside the raster kernels (HIP render_simple only)
        self.render_kwargs = {"fused_activations": True} if fused_activations else {}
        self.bg = torch.tensor([0.55717, 0.52256, 0.51045], dtype=torch.float32, device=self.device)
        self.gaussians = GaussianModel(0)

    @property
    def world(self):
        return self.dist.get_world_size(self.pg) if self.dist else 1

    @property
    def rank(self):
        return self.dist.get_rank(self.pg) if self.dist else 0

    def broadcast_parameters(self):
        if self.dist:
            for p in self.params:
                self.dist.broadcast(p.data, src=0, group=self.pg)

    def allreduce_gradients(self):
        """ONE flat fp32 all-reduce over every parameter that has a gradient (identical set on all ranks)."""
        if not self.dist or self.world == 1:
            return 0
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.pg)
        flat /= self.world
        torch.nan_to_num(flat, nan=0.0, posinf=1e5, neginf=-1e5, out=flat)
        off = 0
        for p, g in zip(self.params, grads):
            n = g.numel()
            p.grad = flat[off:off + n].view_as(p).clone() if p.grad is None else p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        return flat.numel() * 4

    def local_loss(self, batch: SceneBatch):
        """Decoder + raster forward for the local scenes; returns the mean loss over them."""
        total = 0.0
        B = batch.positions.shape[0]
        scene_ids = batch.scene_id.tolist()
        attrs = None
        if self.fused_decoder:   # all local scenes through one decoder launch
            from .fused_decoder import split_attrs
            attrs = split_attrs(self.decoder_fwd.forward_scenes(
                [self.planes * self.latents[s][None, :, None, None] for s in scene_ids], batch.positions))
        for b in range(B):
            gs = self.gaussians
            if attrs is not None:
                gs._xyz, gs._scaling, gs._rotation, gs._opacity, color = attrs[b]
                gs._features_dc = color.unsqueeze(1)
            else:
                planes = self.planes * self.latents[scene_ids[b]][None, :, None, None]
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
            loss = (fused_image_loss if self.fused_loss else image_loss_torch)(image, target, **self.loss_w)[0]
            total = total + loss
        return total / B

    def step(self, batch: SceneBatch) -> float:
        self.optim.zero_grad(set_to_none=True)
        loss = self.local_loss(batch)
        loss.backward()
        self.allreduce_gradients()
        self.optim.step()
        return float(loss.detach())
