"""Generate tests/golden/sequential_decoder_fixture.npz by running the REFERENCE's own
main/decoder_models/sequential_decoder_reverse.py:SequentialDecoderReverse (imported from /root/reference, CPU) on
seeded inputs.  Run in the build container:  python tests/golden/make_decoder_golden.py

The class pulls its feature planes from a GAN generator `G` (G.mapping / G.synthesis, :40-41).  The generators are out
of scope; what the fixture pins is everything AFTER the planes -- the tri-plane sampling call
(:42-57), the five chained heads with their concatenation order (:66-85), activate_scale (:35-36) and the
`* 0.01 + init_position` epilogue (:84) -- so `G` is a 10-line stand-in that returns a seeded [1,3,32,R,R] plane tensor
and the eg3d plane axes; every line of the decoder itself is the reference's.  Only the resulting arrays travel.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)                                  # `main.decoder_models...`
sys.path.insert(0, os.path.join(REF, "eg3d"))            # dnnlib, torch_utils, training.volumetric_rendering

from training.volumetric_rendering.renderer import generate_planes   # noqa: E402
from main.decoder_models.sequential_decoder_reverse import SequentialDecoderReverse  # noqa: E402
from main.decoder_models.sequential_decoder import SequentialDecoder  # noqa: E402
from main.decoder_models.parallel_decoder import ParallelDecoder  # noqa: E402


class StubG(torch.nn.Module):
    def __init__(self, planes):
        super().__init__()
        self.planes = planes
        self.rendering_kwargs = {"box_warp": 1}
        self.renderer = types.SimpleNamespace(plane_axes=generate_planes())

    def mapping(self, z, c, truncation_psi=1.0):
        return z

    def synthesis(self, ws, c, noise_mode="const"):
        return {"feature_planes": self.planes}


def main():
    g = torch.Generator().manual_seed(31)
    planes = torch.randn(1, 3, 32, 24, 24, generator=g)
    pos = torch.rand(96, 3, generator=g) - 0.5
    torch.manual_seed(32)
    dec = SequentialDecoderReverse(StubG(planes), hidden_dim=128, use_xyz_embedding=False, use_gen_finetune=False,
                                   device="cpu")
    dec.triplane_sr = "None"   # set by the training script (train_pano2gaussian_decoder.py), read at :58
    # the default nn.Linear init gives outputs ~1e-2; scale the weights up so the fixture exercises the GELUs
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if p.dim() == 2:
                p.mul_(1.5)
    with torch.no_grad():
        out = dec(torch.zeros(1, 512), torch.zeros(1, 25), pos, 1.0)
    sd = {"sd_" + k: v.numpy() for k, v in dec.state_dict().items() if "decoder" in k}
    np.savez_compressed(os.path.join(HERE, "sequential_decoder_fixture.npz"), planes=planes[0].numpy(),
                        positions=pos.numpy(), color=out.color.numpy(), opacity=out.opacity.numpy(),
                        rotation=out.rotation.numpy(), scale=out.scale.numpy(), xyz=out.xyz.numpy(), **sd)
    print("wrote sequential_decoder_fixture.npz;", {k: tuple(v.shape) for k, v in out.items()},
          "keys", sorted(sd)[:4], "...")
    # the two other decoder types of main/train_pano2gaussian_decoder.py:170-192 on the same planes / positions
    for cls, tag, seed in ((SequentialDecoder, "forward_chain", 33), (ParallelDecoder, "parallel", 34)):
        torch.manual_seed(seed)
        d2 = cls(StubG(planes), hidden_dim=128, use_xyz_embedding=False, use_gen_finetune=False, device="cpu")
        with torch.no_grad():
            for n, p in d2.named_parameters():
                if p.dim() == 2:
                    p.mul_(1.5)
            o2 = d2(torch.zeros(1, 512), torch.zeros(1, 25), pos, 1.0)
        sd2 = {"sd_" + k: v.numpy() for k, v in d2.state_dict().items() if "decoder" in k}
        np.savez_compressed(os.path.join(HERE, f"{tag}_decoder_fixture.npz"), color=o2.color.numpy(),
                            opacity=o2.opacity.numpy(), rotation=o2.rotation.numpy(), scale=o2.scale.numpy(),
                            xyz=o2.xyz.numpy(), **sd2)
        print(f"wrote {tag}_decoder_fixture.npz")
    # The positional encoding of the positions (use_xyz_embedding; main/decoder_utils/pos_encoding.py).  The reference's
    # decoder classes cannot be CONSTRUCTED with use_xyz_embedding=True in this environment -- their
    # torch_utils.persistence decorator pickles the constructor arguments' object graph and the Embedder holds local
    # closures ("Can't pickle local object 'Embedder.frequency_activation.<locals>.func'") -- so the fixture pins the
    # Embedder itself, with the settings the decoders give it (sequential_decoder_reverse.py:22).
    from main.decoder_utils.pos_encoding import Embedder
    emb = Embedder(include_input=True, input_dims=3, num_freqs=10)(pos)
    np.savez_compressed(os.path.join(HERE, "position_embedding_fixture.npz"), positions=pos.numpy(), embedding=emb.numpy())
    print("wrote position_embedding_fixture.npz; embedding", tuple(emb.shape))

if __name__ == "__main__":
    main()
