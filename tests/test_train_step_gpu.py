"""The COMPOSED decoder training step on the GPU (BASELINE config 3; reference main/train_pano2gaussian_decoder.py:217-265:
decoder -> attribute assignment :223-227 -> CustomCam :231 -> render_simple :232 -> losses :244-261 -> backward :263 ->
Adam :264).

Every piece of the step is parity-tested in isolation elsewhere; here `DecoderTrainer.step` as a whole -- HIP tri-plane
gather, decoder, activation getters, the asynchronous single-call rasterizer forward, the fused HIP image loss, the HIP
rasterizer backward, the flat gradient buffer and Adam, all on one stream in the order PyTorch's autograd replays them --
is checked against the same trainer class on the CPU whose rasterizer is the oracle (tests/_cpu_render.py) and whose loss is
the torch evaluation (tests/_torch_losses.py).  Both trainers are built from the same seeds, so they start from identical
parameters.  A wrong stream order between the forward (which returns while binning and blend are still running), the loss
kernel and the backward would show up here as a gradient mismatch.

Then the config-3 size (4 scenes x 500 k points, 512 x 512): 30 steps with the fp32 PyTorch decoder and with the fused
bf16-MFMA decoder -- finite, the loss goes down, and the two loss curves stay within 5 % of each other.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SMALL = dict(plane_res=32, plane_channels=32, hidden_dim=128, image_size=64, seed=7)
N_POINTS = 2000


def _make(device, fused_decoder=False, fused_activations=False, decoder_precision="bf16", **kw):
    from gaussian_gan_decoder_amd.train import DecoderTrainer
    cfg = dict(SMALL)
    cfg.update(kw)
    if str(device) == "cpu":
        from _cpu_render import render_simple_cpu
        from _torch_losses import image_loss_torch
        tr = DecoderTrainer("cpu", render_fn=render_simple_cpu, loss_fn=image_loss_torch, n_scenes_total=2,
                            backbone_params=3000, perceptual_weight=0.05, perceptual_width_div=16, **cfg)
    else:
        tr = DecoderTrainer(device, n_scenes_total=2, backbone_params=3000, perceptual_weight=0.05,
                            perceptual_width_div=16, fused_decoder=fused_decoder, fused_activations=fused_activations,
                            decoder_precision=decoder_precision, **cfg)
    # splats large enough that the 64 x 64 image sees the 2 000 points (the decoder emits -softplus(s+5)-2.5 ~ -7.5)
    with torch.no_grad():
        tr.decoder.scale_decoder.backbone[-1].bias += 3.5
        tr.decoder.opacity_decoder.backbone[-1].bias += 1.0
    return tr


def _flat(params):
    return torch.cat([p.detach().reshape(-1).cpu() for p in params])


def _half_step(tr, batch):
    """step() up to (not including) the optimizer: returns the loss; the flat gradient is in tr.flat_grad."""
    tr.flat_grad.zero_()
    loss = tr.local_loss(batch)
    loss.backward()
    return float(loss.detach())


# the two plane formats of the reference's generators: EG3D tri-planes, PanoHead tri-grids (depth 3, PanoHead's plane axes:
# main/train_pano2gaussian_decoder.py:43 default "panohead", PanoHead/train.py:230,318)
PLANES = {"eg3d": dict(), "panohead": dict(plane_axes="panohead", triplane_depth=3)}


@pytest.mark.parametrize("planes", ["eg3d", "panohead"])
@pytest.mark.parametrize("fused_activations", [False, True], ids=["torch-getters", "fused-activations"])
def test_composed_step_matches_the_oracle_backed_cpu_trainer(native_lib, fused_activations, planes):
    from gaussian_gan_decoder_amd.train import make_scene_batch
    dev = torch.device("cuda:0")
    cpu_tr, gpu_tr = _make("cpu", **PLANES[planes]), _make(dev, fused_activations=fused_activations, **PLANES[planes])
    assert gpu_tr.fused_planes and not cpu_tr.fused_planes   # HIP: modulation inside the gather; CPU: materialised planes
    assert torch.equal(_flat(cpu_tr.params), _flat(gpu_tr.params)), "the two trainers must start from the same parameters"
    for it in range(2):
        cb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], "cpu", seed=it)
        gb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=it)
        lc = _half_step(cpu_tr, cb)
        lg = _half_step(gpu_tr, gb)
        torch.cuda.synchronize()
        assert abs(lc - lg) <= 1e-5 * max(1.0, abs(lc)), (it, lc, lg)
        # gradients, parameter tensor by parameter tensor: |gpu - cpu| <= 1e-10 + 1e-3 * max|g| of that tensor.  The loss is
        # mean-reduced, so the gradients are 1e-3 .. 1e-8 and an absolute 1e-5 would test nothing; the CPU side is itself
        # an fp32 evaluation (oracle blend + torch decoder, sums in another order), hence the relative part
        gc, gg = cpu_tr.flat_grad.clone(), gpu_tr.flat_grad.detach().cpu()
        assert torch.isfinite(gg).all()
        off, worst = 0, 0.0
        for p in cpu_tr.params:
            n = p.numel()
            a, b = gc[off:off + n], gg[off:off + n]
            scale = float(a.abs().max())
            err = float((a - b).abs().max())
            worst = max(worst, err / (1e-10 + 1e-3 * scale))
            off += n
        assert worst <= 1.0, f"step {it}: gradient mismatch, worst |err| / tol = {worst:.2f}"
        assert float(gc.abs().max()) > 1e-4, "the step has no gradient signal"
        cpu_tr.allreduce_and_step()
        gpu_tr.allreduce_and_step()
        torch.cuda.synchronize()
        pc, pg = _flat(cpu_tr.params), _flat(gpu_tr.params)
        # Adam's first steps move every parameter by ~lr * g / (|g| + 1e-8): parameters after the step within 1e-4
        # (lr = 9e-5, the reference's); and all but a sliver (elements whose gradient is at the 1e-8 level) within 2e-6
        d = (pc - pg).abs()
        assert float(d.max()) <= 1e-4, (it, float(d.max()))
        assert float((d > 2e-6).float().mean()) <= 2e-3, (it, float((d > 2e-6).float().mean()))
    init = _flat(_make("cpu", **PLANES[planes]).params)
    assert float((pg - init).abs().max()) > 1e-5      # the steps trained something


@pytest.mark.parametrize("planes", ["eg3d", "panohead"])
def test_composed_step_with_the_fused_decoder(native_lib, planes):
    """Same composition with the bf16-MFMA decoder kernels in place of the PyTorch modules: the loss within 2 % and the
    gradients within bf16-operand accuracy (relative L2 per parameter tensor) of the oracle-backed CPU trainer."""
    from gaussian_gan_decoder_amd.train import make_scene_batch
    dev = torch.device("cuda:0")
    cpu_tr, gpu_tr = _make("cpu", **PLANES[planes]), _make(dev, fused_decoder=True, fused_activations=True, **PLANES[planes])
    cb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], "cpu", seed=0)
    gb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=0)
    lc, lg = _half_step(cpu_tr, cb), _half_step(gpu_tr, gb)
    torch.cuda.synchronize()
    assert abs(lc - lg) <= 2e-2 * abs(lc), (lc, lg)
    gc, gg = cpu_tr.flat_grad.clone(), gpu_tr.flat_grad.detach().cpu()
    assert torch.isfinite(gg).all()
    off = 0
    for p in cpu_tr.params:
        n = p.numel()
        a, b = gc[off:off + n], gg[off:off + n]
        off += n
        if float(a.norm()) < 1e-7:
            continue
        rel = float((a - b).norm() / a.norm())
        assert rel <= 0.12, (tuple(p.shape), rel)


@pytest.mark.parametrize("planes", ["eg3d", "panohead"])
def test_composed_step_with_the_fused_decoder_at_fp32_precision(native_lib, planes):
    """The composition with the split-operand (reference-precision) decoder kernels: loss within 1e-5 and every parameter
    tensor's gradient within 5e-3 of its largest element of the oracle-backed CPU trainer (whose decoder is the fp32 torch
    module)."""
    from gaussian_gan_decoder_amd.train import make_scene_batch
    dev = torch.device("cuda:0")
    cpu_tr = _make("cpu", **PLANES[planes])
    gpu_tr = _make(dev, fused_decoder=True, fused_activations=True, decoder_precision="fp32", **PLANES[planes])
    cb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], "cpu", seed=0)
    gb = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=0)
    lc, lg = _half_step(cpu_tr, cb), _half_step(gpu_tr, gb)
    torch.cuda.synchronize()
    assert abs(lc - lg) <= 1e-5 * max(1.0, abs(lc)), (lc, lg)
    gc, gg = cpu_tr.flat_grad.clone(), gpu_tr.flat_grad.detach().cpu()
    assert torch.isfinite(gg).all()
    off, worst = 0, 0.0
    for p in cpu_tr.params:
        n = p.numel()
        a, b = gc[off:off + n], gg[off:off + n]
        off += n
        worst = max(worst, float((a - b).abs().max()) / (1e-10 + 5e-3 * float(a.abs().max())))
    assert worst <= 1.0, worst


def _curve(tr, steps, batch_fn):
    out = []
    for it in range(steps):
        out.append(tr.step(batch_fn(it)))
    return np.asarray(out)


@pytest.mark.parametrize("planes", ["panohead", "eg3d"])
def test_config3_size_trains(native_lib, planes):
    """BASELINE config 3 ("full train_pano2gaussian_decoder.py step (PanoHead), batch=4, 512^2"): batch = 4 scenes x 500 000
    points at 512 x 512, L1 + L2 + SSIM + Sobel + the perceptual slot, backward through the raster, Adam -- 30 steps with
    the fp32 PyTorch decoder and with the fused bf16-MFMA decoder on the same 4 scenes (bf16 and split-operand fp32
    precision): finite, the loss decreases, the curves agree within 5 %.  "panohead": the named form, tri-grids
    [3, 96, 256, 256] sampled with the 3-D grid_sample; "eg3d": the tri-plane form beside it."""
    from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
    dev = torch.device("cuda:0")
    steps, B, N, S = 30, 4, 500_000, 512
    batch = make_scene_batch(list(range(B)), N, S, dev, seed=0)
    curves = {}
    for name, kw in (("fp32", dict()), ("fused", dict(fused_decoder=True, fused_activations=True)),
                     ("fused-fp32", dict(fused_decoder=True, fused_activations=True, decoder_precision="fp32"))):
        tr = DecoderTrainer(dev, n_scenes_total=B, image_size=S, seed=11, lr=1e-3, perceptual_weight=0.05,
                            perceptual_width_div=4, backbone_params=100_000, **kw, **PLANES[planes])
        c = _curve(tr, steps, lambda it: batch)
        torch.cuda.synchronize()
        print(f"\n  loss {name:5s}:", np.array2string(c[::3], precision=5))
        assert np.isfinite(c).all(), (name, c)
        assert np.mean(c[-5:]) < 0.995 * np.mean(c[:5]) and c[-1] < c[0], (name, c)
        assert all(torch.isfinite(p).all() for p in tr.params), name
        curves[name] = c
        del tr
        torch.cuda.empty_cache()
    for name in ("fused", "fused-fp32"):
        rel = np.abs(curves[name] - curves["fp32"]) / np.abs(curves["fp32"])
        print(f"  max relative difference of the {name} curve from the fp32 curve: {rel.max():.3%}")
        assert rel.max() <= 0.05, (name, rel)


def test_collective_path_on_rccl_with_one_rank(native_lib):
    """The hook-launched all-reduce path (`force_comm`) on the RCCL backend with ONE rank on the one GPU of the box: communicator
    initialisation, async_op work handles launched from inside the backward, their stream semantics under `_timed_wait`, the
    bounded re-arm -- none of which gloo exercises.  The sum over one rank is the identity, so two steps must leave exactly the
    parameters of a trainer that never saw a process group (up to the run-to-run summation order of float atomics); the payload and the share launched from the backward are checked.
    (Not a scaling measurement: pattern of eg3d/training/training_loop.py:288-299.)"""
    import torch.distributed as dist
    from gaussian_gan_decoder_amd.train import make_scene_batch
    dev = torch.device("cuda:0")
    assert not dist.is_initialized()
    plain = _make(dev, fused_decoder=True, fused_activations=True)
    assert not plain._comm
    batch = make_scene_batch([0, 1], N_POINTS, SMALL["image_size"], dev, seed=0)
    losses_plain = [plain.step(batch) for _ in range(2)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = _make(dev, fused_decoder=True, fused_activations=True, force_comm=True)
        assert tr._comm and tr.world == 1 and len(tr._hooks) == len(tr.params)
        tr.measure_comm = True
        losses = [tr.step(batch) for _ in range(2)]
        torch.cuda.synchronize()
        n_params = sum(p.numel() for p in tr.params)
        assert tr.last_allreduce_bytes == 4 * n_params
        assert tr.last_allreduce_bytes_in_backward >= 0.75 * tr.last_allreduce_bytes
        assert tr.allreduce_exposed_ms >= 0.0
        # (equal up to the summation order of the backward's float atomics, which differs from run to run)
        assert np.allclose(losses, losses_plain, rtol=1e-5, atol=0.0), (losses, losses_plain)
        assert float((_flat(tr.params) - _flat(plain.params)).abs().max()) <= 2e-5
        # a step that fails half way leaves units in flight: the next arm waits for them (bounded) and training goes on
        tr.flat_grad.zero_()
        tr.local_loss(batch).backward()
        assert any(u["work"] is not None for u in tr.units)
        tr._arm_units()
        assert all(u["work"] is None for u in tr.units)
        assert np.isfinite(tr.step(batch))
    finally:
        dist.destroy_process_group()
