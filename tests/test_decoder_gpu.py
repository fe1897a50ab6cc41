"""GPU tests of the decoder-side HIP kernel(s): the tri-plane gather (csrc/ggd_triplane.hip) against the plain PyTorch
fp32 ops it replaces (sample_from_planes -> mean over planes), forward and backward."""
import pytest
import torch

from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, sample_from_planes, triplane_mean

pytestmark = pytest.mark.gpu


# N >= 49152 with C in {16, 32, 64}: the sorted-run backward; smaller N or other C: one atomic per (point, tap, channel)
@pytest.mark.parametrize("C,H,W,N", [(32, 64, 64, 10000), (32, 256, 256, 200001), (8, 16, 24, 777), (64, 32, 32, 4096),
                                     (32, 64, 48, 30011), (16, 40, 40, 60000), (64, 24, 40, 50001), (32, 512, 512, 300000),
                                     (8, 32, 32, 70000)])
def test_triplane_mean_matches_torch(native_lib, C, H, W, N):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + N)
    planes = torch.randn(3, C, H, W, generator=g).to(dev).requires_grad_(True)
    pos = (torch.rand(N, 3, generator=g) * 1.2 - 0.6).to(dev)        # some points fall outside the box (zero padding)
    ref_planes = planes.detach().clone().requires_grad_(True)
    out = triplane_mean(planes, pos, 1.0)
    ref = sample_from_planes(ref_planes, pos, 1.0).mean(0)
    assert out.shape == ref.shape == (N, C)
    assert (out - ref).abs().max().item() <= 1e-5
    gout = torch.randn(N, C, generator=g).to(dev)
    out.backward(gout)
    ref.backward(gout)
    scale = max(1.0, ref_planes.grad.abs().max().item())
    assert (planes.grad - ref_planes.grad).abs().max().item() <= 1e-5 * scale * 10   # fp32 atomics: order-dependent sums


def test_sorted_scatter_with_a_live_tile_count_just_below_a_power_of_four(native_lib):
    """3 N = 1 048 800 items = 257 tiles of the depth-sort kernels, of which 600 items (200 points far outside the box) are
    dropped: 256 LIVE tiles.  A pass sizes its look-back groups from the live count (16 groups of 16) while the status words
    were laid out for the launched count (9 group rows until round 4: the surplus rows overwrote the next pass's tile words and
    the order -- hence the run accumulation of the gradient -- came out wrong)."""
    dev = torch.device("cuda:0")
    N, C, H, W = 349_600, 32, 64, 64
    g = torch.Generator().manual_seed(99)
    planes = torch.randn(3, C, H, W, generator=g).to(dev).requires_grad_(True)
    pos = torch.rand(N, 3, generator=g) * 0.9 - 0.45
    pos[torch.randperm(N, generator=g)[:200]] = 5.0
    pos = pos.to(dev)
    ref_planes = planes.detach().clone().requires_grad_(True)
    out = triplane_mean(planes, pos, 1.0)
    ref = sample_from_planes(ref_planes, pos, 1.0).mean(0)
    assert (out - ref).abs().max().item() <= 1e-5
    gout = torch.randn(N, C, generator=g).to(dev)
    out.backward(gout)
    ref.backward(gout)
    scale = max(1.0, ref_planes.grad.abs().max().item())
    assert (planes.grad - ref_planes.grad).abs().max().item() <= 1e-4 * scale


def _trigrid_reference64(planes, pos, axes, D, mod=None):
    """float64 autograd through torch's own 3-D grid_sample on the CPU: features and d(sum(features * gout)) / d planes."""
    p64 = planes.detach().double().cpu().requires_grad_(True)
    src = p64 if mod is None else p64 * mod.detach().double().cpu()[None, :, None, None]
    return p64, sample_from_planes(src, pos.detach().double().cpu(), 1.0, axes, D).mean(0)


@pytest.mark.parametrize("C,D,H,W,N,axes", [(32, 3, 256, 256, 500_000, "panohead"),   # the config-3 tri-grid: sorted runs
                                            (32, 3, 64, 64, 60_000, "eg3d"), (16, 2, 40, 24, 70_001, "panohead"),
                                            (64, 1, 32, 32, 50_000, "panohead"), (32, 5, 20, 28, 49_152, "panohead"),
                                            (32, 3, 64, 64, 3_000, "panohead"), (8, 4, 16, 16, 60_000, "eg3d")])   # plain atomics
def test_trigrid_gather_and_scatter_match_grid_sample_in_float64(native_lib, C, D, H, W, N, axes):
    """PanoHead's 3-D grid_sample over the C x D tri-grids (PanoHead/training/volumetric_rendering/renderer.py:47-58),
    forward and backward, against (a) torch's own fp32 grid_sample + autograd on the GPU -- same fp32 texel coordinates, so
    only the summation order differs: features 1e-5, plane gradients 1e-5 of the largest element -- and (b) the same ops in
    float64 on the CPU: an fp32 texel coordinate of magnitude ~W carries W * 2^-24 of rounding, which a unit step between
    texels turns into ~1e-5 at W = 256: 1e-4."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C * 1000 + D * 100 + N % 97)
    planes = torch.randn(3, C * D, H, W, generator=g)
    # a head-like shell plus points outside the box (zero padding) and points exactly on texel centres / cell borders
    d = torch.randn(N, 3, generator=g)
    pos = 0.3 * d / d.norm(dim=1, keepdim=True) * (1.0 + 0.1 * torch.randn(N, 1, generator=g))
    pos[: N // 50] = torch.rand(N // 50, 3, generator=g) * 1.3 - 0.65
    pos[N // 50: N // 25] = ((torch.randint(0, W, (N // 25 - N // 50, 3), generator=g).float() + 0.5) / W) - 0.5
    gout = torch.randn(N, C, generator=g)
    p64, ref = _trigrid_reference64(planes, pos, axes, D)
    (ref * gout.double()).sum().backward()
    p32 = planes.to(dev).requires_grad_(True)
    ref32 = sample_from_planes(p32, pos.to(dev), 1.0, axes, D).mean(0)
    ref32.backward(gout.to(dev))
    pg = planes.to(dev).requires_grad_(True)
    out = triplane_mean(pg, pos.to(dev), 1.0, axes, D)
    assert out.shape == (N, C)
    assert (out.detach() - ref32.detach()).abs().max().item() <= 1e-5
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    out.backward(gout.to(dev))
    gmax = max(1.0, p64.grad.abs().max().item())
    assert (pg.grad - p32.grad).abs().max().item() <= 1e-5 * gmax
    assert (pg.grad.cpu().double() - p64.grad).abs().max().item() <= 1e-4 * gmax
    assert gmax > 1.0


@pytest.mark.parametrize("C,D,H,W,N,B,axes", [(32, 3, 128, 128, 120_000, 3, "panohead"), (32, None, 128, 128, 100_000, 2, "eg3d"),
                                              (16, 2, 32, 32, 5_000, 2, "panohead")])
def test_modulated_scene_batch_gather_matches_materialised_planes(native_lib, C, D, H, W, N, B, axes):
    """planes_gather on ONE channel-last copy of shared planes with per-scene modulations (the training step's fused form)
    == gathering from the materialised `planes * code` of every scene: features and the gradient w.r.t. the shared planes
    (all scenes summed into one buffer), against float64."""
    from gaussian_gan_decoder_amd.decoder import planes_channels_last, planes_gather
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5 + N)
    depth = D or 1
    planes = torch.randn(3, C * depth, H, W, generator=g)
    codes = 1.0 + 0.25 * torch.randn(B, C * depth, generator=g)
    pos = torch.rand(B, N, 3, generator=g) * 0.9 - 0.45
    gout = torch.randn(B * N, C, generator=g)
    p64 = planes.double().requires_grad_(True)
    ref = torch.cat([sample_from_planes(p64 * codes[b].double()[None, :, None, None], pos[b].double(), 1.0, axes, D).mean(0)
                     for b in range(B)])
    (ref * gout.double()).sum().backward()
    pg = planes.to(dev).requires_grad_(True)
    mods = codes.to(dev).view(B, C, depth).transpose(1, 2).contiguous()
    out = planes_gather(planes_channels_last(pg, D), pos.to(dev), 1.0, axes, D, mod=mods)
    # fp32 texel coordinates against float64 ones: W * 2^-24 of coordinate rounding -> 1e-4 (see the test above); the same
    # computation with materialised planes in fp32 on the GPU: summation order only
    p32 = planes.to(dev).requires_grad_(True)
    ref32 = torch.cat([sample_from_planes(p32 * codes[b].to(dev)[None, :, None, None], pos[b].to(dev), 1.0, axes, D).mean(0)
                       for b in range(B)])
    ref32.backward(gout.to(dev))
    assert (out.detach() - ref32.detach()).abs().max().item() <= 1e-5
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    out.backward(gout.to(dev))
    gmax = max(1.0, p64.grad.abs().max().item())
    assert (pg.grad - p32.grad).abs().max().item() <= 1e-5 * gmax
    assert (pg.grad.cpu().double() - p64.grad).abs().max().item() <= 1e-4 * gmax


def test_decoder_forward_backward_runs_on_gpu(native_lib):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = SequentialDecoderReverse().to(dev)
    planes = torch.randn(3, 32, 64, 64, device=dev, requires_grad=True)
    pos = torch.rand(20000, 3, device=dev) - 0.5
    out = dec(planes, pos)
    loss = out.xyz.sum() + out.scale.sum() + out.rotation.sum() + out.opacity.sum() + out.color.sum()
    loss.backward()
    assert torch.isfinite(planes.grad).all() and planes.grad.abs().sum() > 0
    # same module, torch-only gather (CPU) gives the same outputs
    dec_cpu = SequentialDecoderReverse(); dec_cpu.load_state_dict(dec.state_dict())
    out_cpu = dec_cpu(planes.detach().cpu(), pos.cpu())
    assert (out.xyz.detach().cpu() - out_cpu.xyz).abs().max().item() <= 1e-4


def _f16_reference(dec, feats, pos):
    """PyTorch emulation of the forward kernel's numerics: every layer input and weight rounded to f16, fp32 accumulate;
    the pre-activation rounded to f16 before an exact GELU (the kernel's GELU is a polynomial in packed f16)."""
    r = lambda t: t.to(torch.float16).float()

    def mlp(head, x):
        for k in (0, 2, 4):
            lin = head.backbone[k]
            x = torch.nn.functional.gelu(r(r(x) @ r(lin.weight).t() + lin.bias))
        lin = head.backbone[6]
        return r(x) @ r(lin.weight).t() + lin.bias
    info = pos
    color = mlp(dec.color_decoder, torch.cat([feats, info], 1)); info = torch.cat([info, color], 1)
    opac = mlp(dec.opacity_decoder, torch.cat([feats, info], 1)); info = torch.cat([info, opac], 1)
    rot = mlp(dec.rotation_decoder, torch.cat([feats, info], 1)); info = torch.cat([info, rot], 1)
    scale = dec.activate_scale(mlp(dec.scale_decoder, torch.cat([feats, info], 1))); info = torch.cat([info, scale], 1)
    xyz = mlp(dec.xyz_decoder, torch.cat([feats, info], 1)) * 0.01 + pos
    return dict(color=color, opacity=opac, rotation=rot, scale=scale, xyz=xyz)


@pytest.mark.parametrize("N", [1, 33, 5000, 100003, 1_000_000])   # the last: BASELINE config 4's size
def test_fused_decoder_matches_torch(native_lib, N):
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    dec = SequentialDecoderReverse().to(dev)
    for p in dec.parameters():            # make the heads' outputs O(1) so the comparison is meaningful
        if p.dim() == 2:
            p.data *= 1.5
    planes = torch.randn(3, 32, 64, 64, device=dev)
    pos = torch.rand(N, 3, device=dev) - 0.5
    fused = FusedDecoder(dec)
    out = fused(planes, pos)
    feats = triplane_mean(planes, pos, 1.0)
    with torch.no_grad():
        emu = _f16_reference(dec, feats, pos)
        ref = dec(planes, pos)
    for name in ("color", "opacity", "rotation", "scale", "xyz"):
        got = getattr(out, name)
        assert got.shape == emu[name].shape, name
        # vs the f16 emulation (same rounding points; differences = accumulation order + the GELU polynomial, <= 1.7e-3 per
        # activation, mean 5e-5)
        assert (got - emu[name]).abs().max().item() <= 1e-3 * max(1.0, emu[name].abs().max().item()), name
        # vs the fp32 module: f16 operands (11 significant bits), 4 layers deep.  Measured 3.6e-4 max / 3.5e-5 mean at
        # 1 M points (the bf16 forward of rounds 1-3: 1.2e-3 / 1.4e-4, bound 5e-2)
        assert (got - getattr(ref, name)).abs().max().item() <= 2e-3 * max(1.0, getattr(ref, name).abs().max().item()), name


def test_fused_decoder_training_gradients(native_lib):
    """FusedTrainDecoder (f16-MFMA forward, bf16-MFMA activation backward, split-K weight gradients) vs PyTorch autograd of
    the fp32 module: outputs within the f16 forward's bound, every gradient within bf16-level tolerance."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ref = SequentialDecoderReverse().to(dev)
    for p in ref.parameters():
        if p.dim() == 2:
            p.data *= 1.5
    fused_mod = SequentialDecoderReverse().to(dev)
    fused_mod.load_state_dict(ref.state_dict())
    fused = FusedTrainDecoder(fused_mod)
    N = 20011
    planes_a = torch.randn(3, 32, 64, 64, device=dev, requires_grad=True)
    planes_b = planes_a.detach().clone().requires_grad_(True)
    pos = torch.rand(N, 3, device=dev) - 0.5
    g = torch.Generator().manual_seed(4)
    w = {k: torch.randn(N, d, generator=g).to(dev) for k, d in (("color", 3), ("opacity", 1), ("rotation", 4), ("scale", 3), ("xyz", 3))}

    def loss(o):
        return sum((getattr(o, k) * w[k]).sum() for k in w) / N
    oa = ref(planes_a, pos); loss(oa).backward()
    ob = fused(planes_b, pos); loss(ob).backward()
    for k in w:
        a, b = getattr(oa, k), getattr(ob, k)
        assert (a - b).abs().max().item() <= 2e-3 * max(1.0, a.abs().max().item()), k

    def rel(a, b):
        return ((a - b).norm() / (a.norm() + 1e-12)).item()
    # bf16 activation gradients (dz) bound these; measured 4.6e-3 (planes) and <= 5.7e-3 (every parameter tensor) with the f16
    # forward and the f16 z plane of round 4
    assert rel(planes_a.grad, planes_b.grad) <= 1.5e-2
    for (na, pa), (nb, pb) in zip(ref.named_parameters(), fused_mod.named_parameters()):
        assert pb.grad is not None, nb
        assert rel(pa.grad, pb.grad) <= 1.5e-2, (na, rel(pa.grad, pb.grad))


def test_fused_decoder_matches_reference_class_fixture(native_lib):
    """The fused f16-MFMA decoder against the outputs of the reference's own SequentialDecoderReverse
    (tests/golden/sequential_decoder_fixture.npz, fp32): f16 operand rounding bounds the error (2e-3 of the output
    scale); the fp32 PyTorch path on the GPU must match it to 2e-5."""
    import os
    import numpy as np
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sequential_decoder_fixture.npz"))
    dev = torch.device("cuda:0")
    dec = SequentialDecoderReverse()
    dec.load_state_dict({k[len("sd_"):]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")}, strict=False)
    dec = dec.to(dev)
    planes, pos = torch.from_numpy(f["planes"]).to(dev), torch.from_numpy(f["positions"]).to(dev)
    with torch.no_grad():
        o32 = dec(planes, pos)
        o16 = FusedDecoder(dec)(planes, pos)
    for k in ("color", "opacity", "rotation", "scale", "xyz"):
        ref = f[k]
        np.testing.assert_allclose(getattr(o32, k).cpu().numpy(), ref, atol=2e-5, rtol=1e-5, err_msg=k)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(getattr(o16, k).cpu().numpy() - ref).max()) <= 2e-3 * scale, k


def test_device_pack_matches_the_host_statement_of_the_format(native_lib):
    """ggd_decoder_pack (one launch, what training runs after every optimizer step) == fused_decoder.pack_weights /
    pack_weights_t (torch ops: the host-side statement of the two image formats), byte for byte."""
    from gaussian_gan_decoder_amd.fused_decoder import pack_weights, pack_weights_t, device_pack
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    dec = SequentialDecoderReverse().to(dev)
    for p in dec.parameters():
        p.data = torch.randn_like(p) * 0.7
    packed, packed_t = device_pack(dec)
    assert torch.equal(packed, pack_weights(dec))
    assert torch.equal(packed_t, pack_weights_t(dec))


def test_fused_train_decoder_follows_an_optimizer_step(native_lib):
    """torch's fused Adam updates parameters without bumping Tensor._version; the fused decoder must still decode with
    the new weights (it used to cache its weight images on the version counters and train a frozen decoder)."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    mod = SequentialDecoderReverse().to(dev)
    fused = FusedTrainDecoder(mod)
    planes = torch.randn(3, 32, 32, 32, device=dev)
    pos = torch.rand(4096, 3, device=dev) - 0.5
    opt = torch.optim.Adam(mod.parameters(), lr=1e-2, fused=True)
    outs = []
    for _ in range(3):
        o = fused(planes, pos)
        outs.append(o.color.detach().clone())
        (o.color.square().mean() + o.scale.mean()).backward()
        opt.step(); opt.zero_grad()
    with torch.no_grad():
        ref = mod(planes, pos).color
        now = fused(planes, pos).color
    assert (outs[1] - outs[0]).abs().max().item() > 1e-3 and (outs[2] - outs[1]).abs().max().item() > 1e-3
    assert (now - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N", [1, 33, 5000, 100003, 1_000_000])
def test_fused_decoder_fp32_precision_matches_the_fp32_module(native_lib, N):
    """precision="fp32" (split bf16 operands, three MFMAs per product, csrc/ggd_mlp_hl.inc): every output within 1e-4 of the
    fp32 PyTorch module (the f16 fast form is held to 2e-3) -- the reference trains and evaluates its decoder in fp32
    (main/decoder_models/base_decoder.py:8-27)."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    dec = SequentialDecoderReverse().to(dev)
    for p in dec.parameters():            # make the heads' outputs O(1) so the comparison is meaningful
        if p.dim() == 2:
            p.data *= 1.5
    planes = torch.randn(3, 32, 64, 64, device=dev)
    pos = torch.rand(N, 3, device=dev) - 0.5
    out = FusedDecoder(dec, precision="fp32")(planes, pos)
    with torch.no_grad():
        ref = dec.double()(planes.double(), pos.double())
    worst = 0.0
    for name in ("color", "opacity", "rotation", "scale", "xyz"):
        got, want = getattr(out, name), getattr(ref, name)
        assert got.shape == want.shape, name
        err = (got.double() - want).abs().max().item() / max(1.0, want.abs().max().item())
        worst = max(worst, err)
        assert err <= 1e-4, (name, err)
    print(f"\n  N = {N}: max scaled |fused fp32 - float64 module| = {worst:.2e}")


def test_fused_decoder_fp32_precision_matches_reference_class_fixture(native_lib):
    """precision="fp32" against the outputs of the reference's own SequentialDecoderReverse (fp32; sequential_decoder_fixture.npz):
    1e-4 (the fp32 PyTorch path on the GPU is held to 2e-5, the f16 fast kernel to 2e-3)."""
    import os
    import numpy as np
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sequential_decoder_fixture.npz"))
    dev = torch.device("cuda:0")
    dec = SequentialDecoderReverse()
    dec.load_state_dict({k[len("sd_"):]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")}, strict=False)
    dec = dec.to(dev)
    planes, pos = torch.from_numpy(f["planes"]).to(dev), torch.from_numpy(f["positions"]).to(dev)
    with torch.no_grad():
        o = FusedDecoder(dec, precision="fp32")(planes, pos)
    for k in ("color", "opacity", "rotation", "scale", "xyz"):
        ref = f[k]
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(getattr(o, k).cpu().numpy() - ref).max()) <= 1e-4 * scale, k


def test_fused_decoder_fp32_precision_training_gradients(native_lib):
    """FusedTrainDecoder(precision="fp32") vs float64 autograd of the same module: outputs within 1e-4, every parameter
    gradient and the plane gradient within 1e-3 relative L2 (the bf16 form: 5e-2 / 6e-2)."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ref = SequentialDecoderReverse().to(dev)
    for p in ref.parameters():
        if p.dim() == 2:
            p.data *= 1.5
    fused_mod = SequentialDecoderReverse().to(dev)
    fused_mod.load_state_dict(ref.state_dict())
    ref = ref.double()
    fused = FusedTrainDecoder(fused_mod, precision="fp32")
    N = 200_003
    planes_a = torch.randn(3, 32, 64, 64, device=dev, dtype=torch.float64, requires_grad=True)
    planes_b = planes_a.detach().float().requires_grad_(True)
    pos = torch.rand(N, 3, device=dev) - 0.5
    g = torch.Generator().manual_seed(4)
    w = {k: torch.randn(N, d, generator=g).to(dev) for k, d in (("color", 3), ("opacity", 1), ("rotation", 4), ("scale", 3), ("xyz", 3))}

    def loss(o):
        return sum((getattr(o, k) * w[k].to(getattr(o, k).dtype)).sum() for k in w) / N
    oa = ref(planes_a, pos.double()); loss(oa).backward()
    ob = fused(planes_b, pos); loss(ob).backward()
    for k in w:
        a, b = getattr(oa, k), getattr(ob, k).double()
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item()), k

    def rel(a, b):
        return ((a - b.double()).norm() / (a.norm() + 1e-30)).item()
    report = [("planes", rel(planes_a.grad, planes_b.grad))]
    for (na, pa), (nb, pb) in zip(ref.named_parameters(), fused_mod.named_parameters()):
        assert pb.grad is not None, nb
        report.append((na, rel(pa.grad, pb.grad)))
    print("\n  relative L2 error of the gradients: max %.2e (%s), planes %.2e" % (max(r for _, r in report), max(report, key=lambda t: t[1])[0], report[0][1]))
    for name, r in report:
        assert r <= 1e-3, (name, r)


@pytest.mark.parametrize("loss_scale", [1e-9, 1.0, 3e4])
def test_fused_decoder_fp32_precision_gradients_do_not_depend_on_the_loss_scale(native_lib, loss_scale):
    """The reference-precision kernels keep dz as ONE fp16 plane, every (head, 32-point slab) scaled by its own power of two
    taken from the slab's largest gradient at that head's output (csrc/ggd_mlp_hl.inc): gradients of a loss as small as a mean-reduced image loss (1e-9 per point) or as large as
    3e4 per point must come out with the same RELATIVE accuracy (<= 1e-3 relative L2 against float64 autograd) -- no
    underflow to zero, no overflow to inf.  The per-point weights span four decades (occluded points get tiny gradients next
    to the visible ones', as in a rendered scene)."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ref = SequentialDecoderReverse().to(dev)
    fused_mod = SequentialDecoderReverse().to(dev)
    fused_mod.load_state_dict(ref.state_dict())
    ref = ref.double()
    fused = FusedTrainDecoder(fused_mod, precision="fp32")
    N = 60_000
    planes_a = torch.randn(3, 32, 64, 64, device=dev, dtype=torch.float64, requires_grad=True)
    planes_b = planes_a.detach().float().requires_grad_(True)
    pos = torch.rand(N, 3, device=dev) - 0.5
    g = torch.Generator().manual_seed(6)
    decade = 10.0 ** (-4.0 * torch.rand(N, 1, generator=g))
    w = {k: (torch.randn(N, d, generator=g) * decade * loss_scale).to(dev)
         for k, d in (("color", 3), ("opacity", 1), ("rotation", 4), ("scale", 3), ("xyz", 3))}

    def loss(o):
        return sum((getattr(o, k) * w[k].to(getattr(o, k).dtype)).sum() for k in w)
    loss(ref(planes_a, pos.double())).backward()
    loss(fused(planes_b, pos)).backward()
    worst = ("planes", ((planes_a.grad - planes_b.grad.double()).norm() / planes_a.grad.norm()).item())
    for (na, pa), (nb, pb) in zip(ref.named_parameters(), fused_mod.named_parameters()):
        assert torch.isfinite(pb.grad).all(), nb
        r = ((pa.grad - pb.grad.double()).norm() / (pa.grad.norm() + 1e-300)).item()
        worst = max(worst, (na, r), key=lambda t: t[1])
    print(f"\n  loss scale {loss_scale:g}: worst relative L2 gradient error {worst[1]:.2e} ({worst[0]})")
    assert worst[1] <= 1e-3, worst


def test_fused_decoder_fp32_precision_survives_pre_activations_beyond_fp16(native_lib):
    """z is kept as one fp16 plane: a pre-activation beyond +-65504 is clamped there when it is stored (gelu' is 0 / 1 on
    both sides of the clamp), so the gradients stay finite and the affected rows' inputs still get gradients of the
    right size.  (Stored unclamped it became inf, gelu(inf) * 0 = NaN, and the trainer's nan_to_num silently zeroed whole
    weight-gradient elements.)"""
    from gaussian_gan_decoder_amd.fused_decoder import FusedTrainDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    mod = SequentialDecoderReverse().to(dev)
    with torch.no_grad():
        mod.color_decoder.backbone[0].weight *= 4.0e4     # first-layer pre-activations of the colour head ~ 1e5
    ref = SequentialDecoderReverse().to(dev)
    ref.load_state_dict(mod.state_dict())
    ref = ref.double()
    fused = FusedTrainDecoder(mod, precision="fp32")
    N = 20_000
    planes_a = torch.randn(3, 32, 32, 32, device=dev, dtype=torch.float64, requires_grad=True)
    planes_b = planes_a.detach().float().requires_grad_(True)
    pos = torch.rand(N, 3, device=dev) - 0.5
    oa, ob = ref(planes_a, pos.double()), fused(planes_b, pos)
    (oa.color.sum() + oa.xyz.sum()).backward()
    (ob.color.sum() + ob.xyz.sum()).backward()
    for (na, pa), (nb, pb) in zip(ref.named_parameters(), mod.named_parameters()):
        assert torch.isfinite(pb.grad).all(), nb
        if pa.grad.norm() > 0:
            r = ((pa.grad - pb.grad.double()).norm() / pa.grad.norm()).item()
            assert r <= 5e-3, (na, r)
    assert torch.isfinite(planes_b.grad).all()


def test_split_attrs_on_the_gpu_matches_slicing(native_lib):
    """fused_decoder.split_attrs on CUDA tensors (one HIP pass per scene each way: ggd_attrs_split / ggd_attrs_merge) == the
    slices it replaces, values and gradients, including an output that receives no gradient (zeros in its columns)."""
    from gaussian_gan_decoder_amd import fused_decoder as FD
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = torch.randn(3, 1001, 16, device=dev, requires_grad=True)
    parts = FD.split_attrs(a)
    for b in range(3):
        for t, (lo, hi) in zip(parts[b], FD._SplitAttrs.COLS):
            assert t.is_contiguous() and torch.equal(t, a[b, :, lo:hi])
    w = [[torch.randn_like(t) for t in scene] for scene in parts]
    sum((t * wt).sum() for scene, ws in zip(parts, w) for t, wt in zip(scene, ws) if t.shape[1] != 4).backward()
    ref = torch.zeros_like(a)
    for b in range(3):
        for (lo, hi), wt in zip(FD._SplitAttrs.COLS, w[b]):
            if hi - lo != 4:
                ref[b, :, lo:hi] = wt
    assert torch.equal(a.grad, ref)


def test_exploding_preactivation_stays_inside_its_point(native_lib):
    """The f16 tier's packed GELU does not saturate (csrc/ggd_mlp.hip: z > 65504 -> +inf, z < -131008 -> NaN; the
    reference-precision tier clamps).  Pinned here: a handful of points whose plane features drive the colour head's first
    layer far beyond the f16 range may come out non-finite in THEIR attributes, every other point equals the run without them
    bit for bit, and the fp32 tier stays finite everywhere."""
    from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    dec = SequentialDecoderReverse().to(dev)
    with torch.no_grad():
        dec.color_decoder.backbone[0].weight *= 60.0
    N = 4096
    feats = torch.randn(N, 32, device=dev)
    pos = torch.rand(N, 3, device=dev) - 0.5
    bad = torch.tensor([5, 77, 1000, 4095], device=dev)
    hot = feats.clone()
    hot[bad] *= 6.0e4                                        # clamped to +-65504 on load: pre-activations ~ +-1e6
    f16, f32 = FusedDecoder(dec), FusedDecoder(dec, precision="fp32")
    base, out, out32 = f16.decode_features(feats, pos), f16.decode_features(hot, pos), f32.decode_features(hot, pos)
    keep = torch.ones(N, dtype=torch.bool, device=dev)
    keep[bad] = False
    # attrs rows: [0..2] colour, [3] opacity, [4..7] rotation, [8..10] scale, [11..13] xyz
    assert torch.equal(base[keep, :14], out[keep, :14])      # nobody else is touched
    assert torch.isfinite(out32[:, :14]).all()               # the reference-precision tier saturates
    assert not torch.isfinite(out[bad, :3]).all()            # (the documented behaviour of this tier, not a promise)
