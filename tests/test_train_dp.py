"""CPU tests of the data-parallel decoder training step (world_size 2, gloo) and of the decoder modules.

The raster inside the step is the TEST-ONLY oracle-backed CPU renderer (tests/_cpu_render.py) injected through
DecoderTrainer(render_fn=...); on the GPU the default render_fn is the HIP render_simple (covered by -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussian_gan_decoder_amd.decoder import Decoder, SequentialDecoderReverse, sample_from_planes
from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch

CFG = dict(plane_res=16, plane_channels=8, hidden_dim=16, image_size=32, seed=3)
N_POINTS = 300


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_trainer(n_scenes_total=2):
    """The full step of the bench in miniature: backbone gradient payload (a second all-reduce bucket once it passes
    BUCKET_BYTES; here it shares the bucket), perceptual stand-in, bucketed flat all-reduce + per-bucket Adam."""
    from _cpu_render import render_simple_cpu
    from _torch_losses import image_loss_torch
    tr = DecoderTrainer("cpu", render_fn=render_simple_cpu, loss_fn=image_loss_torch, lr=1e-3, backbone_params=5000,
                        perceptual_weight=0.05, perceptual_width_div=16, n_scenes_total=n_scenes_total, **CFG)
    # larger splats so the 32x32 image actually sees the 300 points
    tr.decoder.scale_decoder.backbone[-1].bias.data += 3.0
    return tr


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import gaussian_gan_decoder_amd.train as T
    T.BUCKET_BYTES = 8192      # several buckets and several all-reduce chunks per bucket even at this toy size
    tr = _make_trainer(world)
    assert len(tr.buckets) >= 2
    losses = []
    total = sum(p.numel() for p in tr.params)
    assert len(tr.units) >= len(tr.buckets) and sum(u["end"] - u["start"] for u in tr.units) == total
    for it in range(2):
        batch = make_scene_batch([rank], N_POINTS, CFG["image_size"], "cpu", seed=it)   # one scene per rank
        # the step, taken apart: the all-reduces must already be in flight when the backward returns (launched by the
        # post-accumulate-grad hooks), and the payload is every parameter exactly once whatever the number of ranks
        tr.flat_grad.zero_()
        loss = tr.local_loss(batch)
        loss.backward()
        in_flight = sum(u["work"] is not None for u in tr.units)
        assert in_flight == len(tr.units), (in_flight, len(tr.units))
        nbytes = tr.allreduce_and_step()
        assert nbytes == 4 * total == tr.last_allreduce_bytes
        losses.append(float(loss.detach()))
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
    torch.save(dict(flat=flat, losses=losses), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_data_parallel_step_matches_single_process(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    r0 = rs[0]
    # replicas stay bit-identical after the all-reduced steps
    for r in rs[1:]:
        assert torch.equal(r0["flat"], r["flat"])
    # and equal the single-process run over the global batch (mean loss == mean of per-rank means)
    tr = _make_trainer(world)
    ref_losses = []
    for it in range(2):
        batch = make_scene_batch(list(range(world)), N_POINTS, CFG["image_size"], "cpu", seed=it)
        ref_losses.append(tr.step(batch))
    ref = torch.cat([p.detach().reshape(-1) for p in tr.params])
    assert torch.allclose(r0["flat"], ref, rtol=1e-4, atol=1e-6), float((r0["flat"] - ref).abs().max())
    # (the backbone stand-in's term is added once per rank, so it appears `world` times in the rank mean)
    bb = float(1e-8 * torch.dot(torch.zeros(1), torch.zeros(1)))
    assert abs(sum(r["losses"][0] for r in rs) / world - ref_losses[0]) < 1e-5 + abs(bb)
    # the step actually trained something
    tr0 = _make_trainer(world)
    init = torch.cat([p.detach().reshape(-1) for p in tr0.params])
    assert (ref - init).abs().max() > 1e-5


def _guard_worker(rank, world, port, out_dir):
    """(a) a second backward before allreduce_and_step() must raise (its gradients would be added to slices whose all-reduce
    is already in flight and never be reduced); (b) after that failed step, step() re-arms the units and the ranks still
    agree; (c) two backward passes under no_sync() + one outside == one backward of the summed loss."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import gaussian_gan_decoder_amd.train as T
    T.BUCKET_BYTES = 8192
    tr = _make_trainer(world)
    batch = make_scene_batch([rank], N_POINTS, CFG["image_size"], "cpu", seed=0)
    # (a)
    tr.flat_grad.zero_()
    tr.local_loss(batch).backward()
    raised = False
    try:
        tr.local_loss(batch).backward()
    except RuntimeError as e:
        raised = "one backward per allreduce_and_step" in str(e)
    assert raised, "a second backward with all-reduces in flight must be refused"
    # (b) the failed step is abandoned; step() waits for what is in flight, re-arms, and trains
    l1 = tr.step(batch)
    assert np.isfinite(l1)
    # (c) accumulation: three micro-batches, the first two under no_sync()
    ref = _make_trainer(world)
    for p, q in zip(ref.params, tr.params):
        p.data.copy_(q.data)
    micro = [make_scene_batch([rank], N_POINTS, CFG["image_size"], "cpu", seed=10 + k) for k in range(3)]
    tr._arm_units(); tr.flat_grad.zero_()
    with tr.no_sync():
        tr.local_loss(micro[0]).backward()
        tr.local_loss(micro[1]).backward()
        assert all(u["work"] is None for u in tr.units), "no collective may start under no_sync()"
        g_acc = tr.flat_grad.clone()                      # (nothing in flight: safe to read)
    tr.local_loss(micro[2]).backward()                    # the pass that completes the accumulation launches the units
    assert all(u["work"] is not None for u in tr.units)
    tr.allreduce_and_step()
    ref._arm_units(); ref.flat_grad.zero_()
    with ref.no_sync():
        (ref.local_loss(micro[0]) + ref.local_loss(micro[1])).backward()
        g_ref = ref.flat_grad.clone()
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
    torch.save(dict(flat=flat, g_acc=g_acc, g_ref=g_ref), os.path.join(out_dir, f"guard{rank}.pt"))
    dist.destroy_process_group()


def test_hooks_refuse_a_second_backward_and_no_sync_accumulates(tmp_path):
    port = _free_port()
    mp.spawn(_guard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"guard{k}.pt") for k in range(2)]
    assert torch.equal(r[0]["flat"], r[1]["flat"])                    # replicas agree after the failed + the accumulated step
    for k in range(2):                                                # two accumulated passes == one backward of the sum
        assert float(r[k]["g_ref"].abs().max()) > 0
        assert torch.allclose(r[k]["g_acc"], r[k]["g_ref"], rtol=1e-4, atol=1e-9)


def test_decoder_shapes_and_param_count():
    d = SequentialDecoderReverse()
    assert sum(p.numel() for p in d.parameters()) == 193294     # "decoder ~0.194 M params"
    assert len(d.get_params_custom()) == 5 * 8
    out = d(torch.randn(3, 32, 16, 16), torch.rand(50, 3) - 0.5)
    assert out.xyz.shape == (50, 3) and out.scale.shape == (50, 3) and out.rotation.shape == (50, 4)
    assert out.opacity.shape == (50, 1) and out.color.shape == (50, 3)
    assert (out.scale <= -2.5).all()                            # -softplus(s+5) - 2.5
    names = [n for n, _ in d.named_parameters()]
    assert "color_decoder.backbone.0.weight" in names and "xyz_decoder.backbone.6.bias" in names


def test_decoder_matches_reference_fixture():
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_fixture.npz"))
    planes = torch.from_numpy(f["planes"]); pos = torch.from_numpy(f["positions"])
    pf = sample_from_planes(planes, pos, box_warp=1.0)
    np.testing.assert_allclose(pf.numpy(), f["plane_features"], atol=1e-6)
    dec = Decoder(int(f["n_features"]), int(f["out_features"]), int(f["hidden_dim"]))
    dec.load_state_dict({k[len("sd_"):]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")})
    out = dec(pf, pos)
    np.testing.assert_allclose(out.detach().numpy(), f["decoder_out"], atol=1e-5, rtol=1e-5)


def test_sequential_decoder_matches_reference_class_fixture():
    """tests/golden/sequential_decoder_fixture.npz: outputs of the REFERENCE's SequentialDecoderReverse
    (main/decoder_models/sequential_decoder_reverse.py, run by tests/golden/make_decoder_golden.py) for seeded planes,
    positions and weights.  Our module must load its state_dict unchanged and reproduce all five heads."""
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sequential_decoder_fixture.npz"))
    dec = SequentialDecoderReverse()
    sd = {k[len("sd_"):]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")}
    missing, unexpected = dec.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "decoder" in m], (missing, unexpected)
    with torch.no_grad():
        out = dec(torch.from_numpy(f["planes"]), torch.from_numpy(f["positions"]))
    for k in ("color", "opacity", "rotation", "scale", "xyz"):
        np.testing.assert_allclose(getattr(out, k).numpy(), f[k], atol=2e-5, rtol=1e-5, err_msg=k)


@pytest.mark.parametrize("tag", ["forward_chain", "parallel"])
def test_other_decoder_types_match_reference_class_fixtures(tag):
    """decoder_type "sequential" / "parallel" of main/train_pano2gaussian_decoder.py:170-192: outputs of the reference's
    SequentialDecoder / ParallelDecoder (main/decoder_models/sequential_decoder.py:38-84, parallel_decoder.py:38-80) on the
    planes / positions of sequential_decoder_fixture.npz; our modules load their state_dict unchanged."""
    from gaussian_gan_decoder_amd.decoder import SequentialDecoder, ParallelDecoder
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    base = np.load(os.path.join(here, "sequential_decoder_fixture.npz"))
    f = np.load(os.path.join(here, f"{tag}_decoder_fixture.npz"))
    dec = (SequentialDecoder if tag == "forward_chain" else ParallelDecoder)()
    missing, unexpected = dec.load_state_dict({k[3:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd_")}, strict=False)
    assert not unexpected and not [m for m in missing if "decoder" in m], (missing, unexpected)
    with torch.no_grad():
        out = dec(torch.from_numpy(base["planes"]), torch.from_numpy(base["positions"]))
    for k in ("color", "opacity", "rotation", "scale", "xyz"):
        np.testing.assert_allclose(getattr(out, k).numpy(), f[k], atol=2e-5, rtol=1e-5, err_msg=k)
    assert (out.scale <= -2.0).all()


def test_position_embedding_matches_the_reference_embedder():
    """use_xyz_embedding: embed_positions == the reference's Embedder(include_input=True, input_dims=3, num_freqs=10)
    (main/decoder_utils/pos_encoding.py) on seeded positions; the three decoder classes accept the 63-column form."""
    from gaussian_gan_decoder_amd.decoder import embed_positions, SequentialDecoderReverse, SequentialDecoder, ParallelDecoder
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "position_embedding_fixture.npz"))
    pos = torch.from_numpy(f["positions"])
    e = embed_positions(pos)
    assert e.shape == (pos.shape[0], 63)
    np.testing.assert_allclose(e.numpy(), f["embedding"], atol=1e-6, rtol=0)
    torch.manual_seed(3)
    planes = torch.randn(3, 32, 16, 16)
    for cls in (SequentialDecoderReverse, SequentialDecoder, ParallelDecoder):
        dec = cls(use_xyz_embedding=True)
        assert dec.color_decoder.backbone[0].in_features in (32 + 63, 32 + 63 + 11)
        out = dec(planes, pos)
        assert out.xyz.shape == (pos.shape[0], 3) and torch.isfinite(out.color).all()
        out.xyz.sum().backward()


def test_allreduce_payload_of_the_full_configuration():
    """BASELINE config 5's payload: decoder 193 294 + shared planes 3 x 32 x 256 x 256 + the backbone stand-in = 29 763 294
    floats (the reference's finetuned generator, sequential_decoder_reverse.py:89-99), i.e. 4 x 29 763 294 bytes per step,
    cut into units of <= 32 MB that cover the flat gradient exactly once -- a property of the trainer, not of the rank count."""
    from _cpu_render import render_simple_cpu
    from _torch_losses import image_loss_torch
    tr = DecoderTrainer("cpu", n_scenes_total=1, render_fn=render_simple_cpu, loss_fn=image_loss_torch,
                        backbone_params=23_278_544)
    total = sum(p.numel() for p in tr.params)
    assert total == 29_763_294
    assert tr.flat_grad.numel() == total
    spans = sorted((u["start"], u["end"]) for u in tr.units)
    assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(e - s for s, e in spans) * 4 <= 32 << 20
    assert sum(e - s for s, e in spans) * 4 == 4 * 29_763_294
    # every parameter belongs to at least one unit, and a unit waits for exactly the parameters that overlap it
    assert all(len(tr._units_of_param[id(p)]) >= 1 for p in tr.params)
    assert sum(u["need"] for u in tr.units) == sum(len(v) for v in tr._units_of_param.values())
    # the same payload in the form BASELINE config 3 / 5 name: PanoHead tri-grids [3, 96, 256, 256], the stand-in holds the rest
    del tr
    tr = DecoderTrainer("cpu", n_scenes_total=1, render_fn=render_simple_cpu, loss_fn=image_loss_torch,
                        plane_axes="panohead", triplane_depth=3, backbone_params=29_570_000 - 3 * 96 * 256 * 256)
    assert tuple(tr.planes.shape) == (3, 96, 256, 256) and tr.decoder.triplane_depth == 3
    assert sum(p.numel() for p in tr.params) == 29_763_294
    assert sum(u["end"] - u["start"] for u in tr.units) == 29_763_294
