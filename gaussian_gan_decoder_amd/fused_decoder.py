"""Inference-time fused decoder: tri-plane gather + the 5 chained MLP heads as ONE MFMA kernel with 16-bit operands
(csrc/ggd_mlp.hip; SURVEY.md section 8f row 1, BASELINE config 3).

`FusedDecoder(decoder)` wraps a `SequentialDecoderReverse` (same parameters; call `.repack()` after they change) and
returns the same namespace (xyz, scale, rotation, opacity, color).  No autograd: training keeps the PyTorch modules.
Numerics of the forward: weights and activations are rounded to f16 (11 significant bits; inputs and weights clamped to
+-65504) at every layer input, accumulation and bias in fp32, GELU as a degree-6 polynomial in packed f16 (<= 1.7e-3, mean
5e-5: scripts/gelu_f16_fit.py).  The training tier named "bf16" runs this forward; its backward kernels keep bf16 operands
(dz needs the exponent range)."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _capi
from .decoder import SequentialDecoderReverse, triplane_mean

HID = 128
ROW1, ROW2 = 64, 128                # 16-bit elements per weight row (no padding; 16-byte slots swizzled, see _swizzle_rows)
# position (g, e) inside a 32-wide k block  <-  k = 4g+e (e < 4) or 16+4g+(e-4): the order the MFMA C/D layout
# hands a layer's outputs to the next layer's B operand (csrc/ggd_mlp.hip header)
_PERM32 = [(4 * g + e) if e < 4 else (16 + 4 * g + (e - 4)) for g in range(4) for e in range(8)]


def _permute_blocks(w: torch.Tensor) -> torch.Tensor:
    K = w.shape[1]
    idx = torch.tensor([32 * s + p for s in range(K // 32) for p in _PERM32], device=w.device)
    return w[:, idx]


_IN_FEATURES = (35, 38, 39, 43, 46)   # colour, opacity, rotation, scale, xyz: 32 plane channels + xyz + the earlier heads


def _check_decoder(decoder) -> None:
    """The kernels hard-code SequentialDecoderReverse's chain (colour -> opacity -> rotation -> scale -> xyz, each head fed
    the earlier heads' outputs) and its scale activation -softplus(s + 5) - 2.5: anything else -- SequentialDecoder (xyz
    first, -2), ParallelDecoder (no chaining), the positional-encoding option -- would pack without complaint (same head
    names, first-layer widths inside 35..48) and then decode silently wrong attributes and gradients."""
    if not isinstance(decoder, SequentialDecoderReverse):
        raise TypeError(f"the fused decoder kernels implement SequentialDecoderReverse only (got {type(decoder).__name__}); "
                        "use the PyTorch module for the other decoder types")
    if getattr(decoder, "use_xyz_embedding", False):
        raise ValueError("the fused decoder kernels do not implement use_xyz_embedding; use the PyTorch module")
    got = tuple(getattr(decoder, n).backbone[0].in_features for n in
                ("color_decoder", "opacity_decoder", "rotation_decoder", "scale_decoder", "xyz_decoder"))
    if got != _IN_FEATURES:
        raise ValueError(f"fused decoder: first-layer widths {got} are not SequentialDecoderReverse's {_IN_FEATURES} "
                         "(32 plane channels, 3-D positions)")


def _swizzle_rows(wp: torch.Tensor) -> torch.Tensor:
    """[rows, K] (K = 64 or 128) -> the same rows with their 8-element (16-byte) slots XOR-swizzled by the row index, the LDS
    image the kernels read bank-conflict free (csrc/ggd_mlp.hip::wslot): physical slot = logical slot ^ (r & 15) for K = 128,
    ^ ((r >> 1) & 7) for K = 64."""
    rows, K = wp.shape
    r = torch.arange(rows, device=wp.device)
    sw = (r & 15) if K == 128 else ((r >> 1) & 7)
    phys = torch.arange(K // 8, device=wp.device)[None, :] ^ sw[:, None]       # [rows, slots]: physical slot of logical slot q
    out = torch.empty_like(wp).view(rows, K // 8, 8)
    out.scatter_(1, phys[:, :, None].expand(rows, K // 8, 8), wp.view(rows, K // 8, 8))
    return out.view(rows, K)


def pack_weights(decoder: SequentialDecoderReverse) -> torch.Tensor:
    """-> uint8 tensor of ggd_decoder_packed_bytes(): per head [W1 128x64 | W2 128x128 | W3 128x128 | W4 16x128] f16 (clamped
    to +-65504, slots swizzled), then b1 b2 b3 [128] and b4 [16] fp32; W1 .. W3 and b1 .. b3 halved."""
    _check_decoder(decoder)
    dev = next(decoder.parameters()).device
    chunks = []
    for head in (decoder.color_decoder, decoder.opacity_decoder, decoder.rotation_decoder, decoder.scale_decoder,
                 decoder.xyz_decoder):
        l1, l2, l3, l4 = head.backbone[0], head.backbone[2], head.backbone[4], head.backbone[6]
        if l1.out_features != HID or l1.in_features < 35 or l1.in_features > 32 + 16 or l4.out_features > 16:
            raise ValueError("fused decoder supports hidden_dim 128, 32 plane channels, <= 13 chained inputs")
        w1 = torch.zeros(HID, 64, device=dev)
        w1[:, :l1.in_features] = l1.weight.detach().float()        # cols 0..31 planes, 32.. = info slots
        w4 = torch.zeros(16, HID, device=dev)
        w4[:l4.out_features] = l4.weight.detach().float()
        b4 = torch.zeros(16, device=dev)
        b4[:l4.out_features] = l4.bias.detach().float()
        rows = []
        # the hidden layers are halved (exact): the forward's accumulators hold z / 2, the form its f16 GELU starts from
        for w, row in ((0.5 * w1, ROW1), (0.5 * l2.weight.detach().float(), ROW2), (0.5 * l3.weight.detach().float(), ROW2),
                       (w4, ROW2)):
            assert row == w.shape[1]
            wp = _swizzle_rows(_permute_blocks(w))
            rows.append(wp.clamp(-65504.0, 65504.0).to(torch.float16).contiguous().view(torch.uint8).reshape(-1))
        biases = torch.cat([0.5 * l1.bias.detach().float(), 0.5 * l2.bias.detach().float(), 0.5 * l3.bias.detach().float(), b4])
        chunks += rows + [biases.contiguous().view(torch.uint8).reshape(-1)]
    packed = torch.cat(chunks).contiguous()
    expect = _capi.load().ggd_decoder_packed_bytes()
    if packed.numel() != expect:
        raise RuntimeError(f"packed decoder image is {packed.numel()} bytes, library expects {expect}")
    return packed


_HEADS = ("color_decoder", "opacity_decoder", "rotation_decoder", "scale_decoder", "xyz_decoder")
_N_EXTRA = (0, 3, 4, 8, 11)   # chained inputs of each head (outputs of the earlier heads)
_OUT_DIM = (3, 1, 4, 3, 3)
ROW4T = 32 + 8


def _head_tensors(decoder):
    """[(W1,b1,W2,b2,W3,b3,W4,b4)] * 5 in head order."""
    _check_decoder(decoder)
    out = []
    for name in _HEADS:
        bb = getattr(decoder, name).backbone
        out.append(tuple(t for k in (0, 2, 4, 6) for t in (bb[k].weight, bb[k].bias)))
    return out


def pack_weights_t(decoder: SequentialDecoderReverse) -> torch.Tensor:
    """Transposed weight image for ggd_decoder_backward: per head [W4^T 128x40 | W3^T 128x128 | W2^T 128x128 |
    W1^T 64x128] bf16, every row's K (= the layer's OUTPUT features) permuted like the forward image; the K = 128 rows
    swizzled like the forward image, W4^T rows padded (32 + 8) and not swizzled."""
    dev = next(decoder.parameters()).device
    chunks = []
    for (w1, _, w2, _, w3, _, w4, _) in _head_tensors(decoder):
        w1f = torch.zeros(HID, 64, device=dev); w1f[:, :w1.shape[1]] = w1.detach().float()
        w4f = torch.zeros(32, HID, device=dev); w4f[:w4.shape[0]] = w4.detach().float()
        for wt, row in ((w4f.t(), ROW4T), (w3.detach().float().t(), ROW2), (w2.detach().float().t(), ROW2),
                        (w1f.t(), ROW2)):
            if row == wt.shape[1]:
                wp = _swizzle_rows(_permute_blocks(wt.contiguous()))
            else:
                wp = torch.zeros(wt.shape[0], row, device=dev)
                wp[:, :wt.shape[1]] = _permute_blocks(wt.contiguous())
            chunks.append(wp.to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1))
    packed = torch.cat(chunks).contiguous()
    expect = _capi.load().ggd_decoder_packed_t_bytes()
    if packed.numel() != expect:
        raise RuntimeError(f"packed transposed image is {packed.numel()} bytes, library expects {expect}")
    return packed


def _splitk_dw(dy: torch.Tensor, x: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
    """dW = dy^T x over N points ([N,out], [N,in] bf16) as a batched split-K product (keeps all CUs busy)."""
    n = dy.shape[0]
    main = (n // chunk) * chunk
    dw = None
    if main:
        dw = torch.bmm(dy[:main].view(-1, chunk, dy.shape[1]).transpose(1, 2),
                       x[:main].view(-1, chunk, x.shape[1])).float().sum(0)
    if main < n:
        tail = (dy[main:].t() @ x[main:]).float()
        dw = tail if dw is None else dw + tail
    return dw


PRECISIONS = ("bf16", "fp32")


def _check_precision(precision: str) -> bool:
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS} (got {precision!r})")
    return precision == "fp32"


def device_pack(decoder, params=None, buffers=None, hl: bool = False):
    """(packed, packed_t) built by the library's pack kernel from the decoder's 40 parameter tensors (head order colour,
    opacity, rotation, scale, xyz; W1 b1 .. W4 b4).  hl=False: the bf16 images -- same bytes as pack_weights /
    pack_weights_t, which stay the host-side statement of that format; hl=True: the split (hi + lo) images of the
    reference-precision kernels (csrc/ggd_mlp_hl.inc).  `buffers`: a previous result to overwrite in place."""
    if params is None:
        params = [t for head in _head_tensors(decoder) for t in head]
    dev = params[0].device
    if dev.type != "cuda":
        raise RuntimeError("the fused decoder is a HIP kernel: CUDA tensors required (use the PyTorch decoder on CPU)")
    ps = [p.detach() for p in params]
    ps = [p if (p.dtype == torch.float32 and p.is_contiguous()) else p.float().contiguous() for p in ps]
    cx = _capi.context_for(dev)
    sizes = ((cx.lib.ggd_decoder_packed_hl_bytes(), cx.lib.ggd_decoder_packed_t_hl_bytes()) if hl else
             (cx.lib.ggd_decoder_packed_bytes(), cx.lib.ggd_decoder_packed_t_bytes()))
    if buffers is None or buffers[0].device != dev or buffers[0].numel() != sizes[0]:
        buffers = (torch.empty((sizes[0],), dtype=torch.uint8, device=dev),
                   torch.empty((sizes[1],), dtype=torch.uint8, device=dev))
    table = (C.c_void_p * 40)(*[p.data_ptr() for p in ps])
    fn = cx.lib.ggd_decoder_pack_hl if hl else cx.lib.ggd_decoder_pack
    with torch.cuda.device(dev):
        cx.check(fn(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), table,
                    C.c_void_p(buffers[0].data_ptr()), C.c_void_p(buffers[1].data_ptr())))
    return buffers


import os as _os

# points per backward / weight-gradient chunk (z + dz of a chunk = 15 KiB per point; 0 = no chunking)
WGRAD_CHUNK = int(_os.environ.get("GGD_WGRAD_CHUNK", "0"))


class FusedDecoderFn(torch.autograd.Function):
    """attrs[N,16] = fused 5-head decoder(feats[N,32], pos[N,3]; 40 weight/bias tensors), differentiable w.r.t. feats
    and the parameters.  Forward and activation-backward are the bf16-MFMA kernels; the weight gradients are split-K
    GEMMs over the kept pre-activations (bf16)."""

    @staticmethod
    def forward(ctx, feats, pos, packed, packed_t, hl, *params):
        dev = feats.device
        feats = feats.contiguous().float()
        pos = pos.contiguous().float()
        n = pos.shape[0]
        cx = _capi.context_for(dev)
        attrs = torch.empty((n, 16), dtype=torch.float32, device=dev)
        # 16-point blocks (ggd_decoder_zbuf_bytes); f16 values in both tiers (opaque here)
        zbuf = torch.empty((5, 3, (n + 15) // 16 * 16, HID), dtype=torch.bfloat16, device=dev)
        fwd = cx.lib.ggd_decoder_forward_hl if hl else cx.lib.ggd_decoder_forward_train
        with torch.cuda.device(dev):
            cx.check(fwd(cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.c_void_p(feats.data_ptr()),
                         C.c_void_p(pos.data_ptr()), n, C.c_void_p(packed.data_ptr()), C.c_void_p(attrs.data_ptr()),
                         C.c_void_p(zbuf.data_ptr())))
        # packed_t is rebuilt IN PLACE by the next forward (FusedTrainDecoder._images), through raw pointers autograd cannot
        # version: the backward of a graph kept across an optimizer step would silently read the new weights.  The node keeps
        # its own copy (0.9 MB, or 1.8 MB in the split form: one small device copy per forward).
        ctx.save_for_backward(feats, pos, attrs, zbuf, packed_t.clone())
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.hl = bool(hl)
        return attrs

    @staticmethod
    def backward(ctx, dattrs):
        feats, pos, attrs, zbuf, packed_t = ctx.saved_tensors
        dev = feats.device
        n = pos.shape[0]
        dattrs = dattrs.contiguous().float()
        cx = _capi.context_for(dev)
        # dz: one 16-bit plane in z's blocked layout (bf16 kernels: bf16 values; reference precision: loss-scaled fp16 values,
        # the scale chosen on the device from max |dattrs| -- csrc/ggd_mlp_hl.inc)
        dzbuf = torch.empty(tuple(zbuf.shape), dtype=torch.bfloat16, device=dev)
        dout = torch.empty((5, n, 4), dtype=torch.float32, device=dev)
        dfeat = torch.empty((n, 32), dtype=torch.float32, device=dev)
        dinfo = torch.empty((n, 16), dtype=torch.float32, device=dev)
        per_head = cx.lib.ggd_decoder_wgrad_floats() // 5
        wg = torch.zeros((5, per_head), dtype=torch.float32, device=dev)
        # activation backward + weight gradients, chunk by chunk (WGRAD_CHUNK points): a chunk's dz / z rows are consumed by
        # the weight-gradient kernel while they are still in the Infinity Cache
        bwd = cx.lib.ggd_decoder_backward_wgrad_hl if ctx.hl else cx.lib.ggd_decoder_backward_wgrad
        with torch.cuda.device(dev):
            cx.check(bwd(
                cx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), n, int(WGRAD_CHUNK),
                C.c_void_p(packed_t.data_ptr()), C.c_void_p(attrs.data_ptr()), C.c_void_p(dattrs.data_ptr()),
                C.c_void_p(zbuf.data_ptr()), C.c_void_p(dzbuf.data_ptr()), C.c_void_p(dout.data_ptr()),
                C.c_void_p(dfeat.data_ptr()), C.c_void_p(dinfo.data_ptr()), C.c_void_p(feats.data_ptr()),
                C.c_void_p(pos.data_ptr()), C.c_void_p(wg.data_ptr())))
        grads = []
        for h in range(5):
            in_dim, od = 35 + _N_EXTRA[h], _OUT_DIM[h]
            o = 0
            for rows, cols, r_used, c_used in ((HID, 64, HID, in_dim), (HID, HID, HID, HID), (HID, HID, HID, HID),
                                               (16, HID, od, HID)):
                w = wg[h, o:o + rows * cols].view(rows, cols)[:r_used, :c_used]
                o += rows * cols
                grads += [w, wg[h, o:o + rows][:r_used]]
                o += rows
        return (dfeat, None, None, None, None, *grads)


class _SplitAttrs(torch.autograd.Function):
    """attrs[B,N,16] -> per scene the five contiguous tensors the rasterizer takes (xyz, scale, rotation, opacity,
    colour).  One autograd node and one HIP pass per scene each way (ggd_attrs_split / ggd_attrs_merge): autograd's own
    chain of select / slice / contiguous nodes costs five strided copies per scene forward, and backward a zero-filled
    full-size tensor plus an add per slice."""
    COLS = ((11, 14), (8, 11), (4, 8), (3, 4), (0, 3))   # xyz, scale, rotation, opacity, colour

    @staticmethod
    def forward(ctx, attrs):
        ctx.shape = attrs.shape
        B, N, _ = attrs.shape
        if not attrs.is_cuda:
            return tuple(attrs[b, :, lo:hi].contiguous() for b in range(B) for lo, hi in _SplitAttrs.COLS)
        attrs = attrs.contiguous().float()
        dev = attrs.device
        cx = _capi.context_for(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        outs = []
        with torch.cuda.device(dev):
            for b in range(B):
                o = [torch.empty((N, hi - lo), dtype=torch.float32, device=dev) for lo, hi in _SplitAttrs.COLS]
                cx.check(cx.lib.ggd_attrs_split(cx.handle, stream, C.c_void_p(attrs[b].data_ptr()), N,
                                                *[C.c_void_p(t.data_ptr()) for t in o]))
                outs += o
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        B, N, _ = ctx.shape
        g0 = next(g for g in grads if g is not None)
        if not g0.is_cuda:
            d = torch.zeros(ctx.shape, dtype=g0.dtype, device=g0.device)
            for k, g in enumerate(grads):
                if g is not None:
                    lo, hi = _SplitAttrs.COLS[k % 5]
                    d[k // 5, :, lo:hi] = g
            return d
        dev = g0.device
        d = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            for b in range(B):
                gs = [None if g is None else g.contiguous().float() for g in grads[5 * b:5 * b + 5]]
                cx.check(cx.lib.ggd_attrs_merge(cx.handle, stream, N, *[None if g is None else C.c_void_p(g.data_ptr()) for g in gs],
                                                C.c_void_p(d[b].data_ptr())))
        return d


def split_attrs(attrs: torch.Tensor):
    """[(xyz, scale, rotation, opacity, colour)] * B from the fused decoder's attrs[B,N,16] (see _SplitAttrs)."""
    flat = _SplitAttrs.apply(attrs)
    return [flat[5 * b:5 * b + 5] for b in range(attrs.shape[0])]


class FusedTrainDecoder(torch.nn.Module):
    """Training front-end with the `SequentialDecoderReverse` call signature: tri-plane gather (HIP) + the fused
    bf16-MFMA decoder with autograd.  Wraps (and shares the parameters of) a SequentialDecoderReverse."""

    def __init__(self, decoder: SequentialDecoderReverse, precision: str = "bf16"):
        """precision: "bf16" = 16-bit operands: f16 forward, bf16 backward (outputs within 2e-3, measured 4e-4; parameter gradients
        within 1.5 % relative L2 of fp32, measured 0.6 %);
        "fp32" = the reference's training precision on the same matrix cores: split bf16 operands, three MFMAs per product
        (outputs within 1e-4, parameter gradients within 1e-3 relative L2 of the fp32 module)."""
        super().__init__()
        self.decoder = decoder
        _check_decoder(decoder)
        self.precision = precision
        self._hl = _check_precision(precision)
        self._image_bufs = None

    def get_params_custom(self):
        return self.decoder.get_params_custom()

    def _images(self, params):
        """Weight images for this call: ONE device launch (ggd_decoder_pack) straight from the parameter tensors, on every
        forward.  (They used to be cached on the parameters' `_version`; torch's fused Adam updates parameters without
        bumping it, so a trainer kept decoding with the images of step 0.)"""
        packed, packed_t = device_pack(self.decoder, params, self._image_bufs, self._hl)
        self._image_bufs = (packed, packed_t)
        return packed, packed_t

    def forward_scenes(self, planes_list, positions, feats=None):
        """Several scenes (own feature planes, positions[B,N,3]) through ONE decoder launch: attrs[B,N,16].  The
        weight gradients are then formed once per step instead of once per scene.  feats [B * N, 32]: the plane-mean
        features when the caller has gathered them already (decoder.planes_gather); planes_list is not read then."""
        B, N = positions.shape[0], positions.shape[1]
        if feats is None:
            feats = torch.cat([triplane_mean(planes_list[b], positions[b], self.decoder.box_warp, self.decoder.plane_axes,
                                             self.decoder.triplane_depth) for b in range(B)], dim=0)
        params = [t for head in _head_tensors(self.decoder) for t in head]
        packed, packed_t = self._images(params)
        a = FusedDecoderFn.apply(feats, positions.reshape(B * N, 3), packed, packed_t, self._hl, *params)
        return a.view(B, N, 16)

    def forward(self, feature_planes, init_position, features=None):
        feats = features if features is not None else \
            triplane_mean(feature_planes, init_position, self.decoder.box_warp, self.decoder.plane_axes,
                          self.decoder.triplane_depth)
        params = [t for head in _head_tensors(self.decoder) for t in head]
        packed, packed_t = self._images(params)
        a = FusedDecoderFn.apply(feats, init_position, packed, packed_t, self._hl, *params)
        return SimpleNamespace(color=a[:, 0:3], opacity=a[:, 3:4], rotation=a[:, 4:8], scale=a[:, 8:11], xyz=a[:, 11:14])


class FusedDecoder:
    def __init__(self, decoder: SequentialDecoderReverse, precision: str = "bf16"):
        self.precision = precision
        self._hl = _check_precision(precision)
        self.decoder = decoder
        self.box_warp = decoder.box_warp
        self.plane_axes, self.triplane_depth = decoder.plane_axes, decoder.triplane_depth
        self.repack()

    def repack(self):
        """Rebuild the weight image after the wrapped decoder's parameters changed."""
        if next(self.decoder.parameters()).is_cuda:
            self.packed = device_pack(self.decoder, hl=self._hl)[0]
        elif self._hl:
            raise RuntimeError("the fused decoder is a HIP kernel: move the decoder to the GPU first")
        else:
            self.packed = pack_weights(self.decoder)

    @torch.no_grad()
    def decode_features(self, feats: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """feats [N,32], positions [N,3] (CUDA fp32) -> attrs [N,16] (layout: include/ggd_raster.h)."""
        if not feats.is_cuda:
            raise RuntimeError("the fused decoder is a HIP kernel: CUDA tensors required (use the PyTorch decoder on CPU)")
        dev = feats.device
        feats = feats.contiguous().float()
        positions = positions.contiguous().float()
        n = positions.shape[0]
        attrs = torch.empty((n, 16), dtype=torch.float32, device=dev)
        cx = _capi.context_for(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            if self._hl:
                cx.check(cx.lib.ggd_decoder_forward_hl(cx.handle, stream, C.c_void_p(feats.data_ptr()),
                                                       C.c_void_p(positions.data_ptr()), n,
                                                       C.c_void_p(self.packed.data_ptr()), C.c_void_p(attrs.data_ptr()), None))
            else:
                cx.check(cx.lib.ggd_decoder_forward(cx.handle, stream, C.c_void_p(feats.data_ptr()),
                                                    C.c_void_p(positions.data_ptr()), n,
                                                    C.c_void_p(self.packed.data_ptr()), C.c_void_p(attrs.data_ptr())))
        return attrs

    @torch.no_grad()
    def __call__(self, feature_planes: torch.Tensor, init_position: torch.Tensor) -> SimpleNamespace:
        feats = triplane_mean(feature_planes, init_position, self.box_warp, self.plane_axes, self.triplane_depth)
        a = self.decode_features(feats, init_position)
        return SimpleNamespace(color=a[:, 0:3], opacity=a[:, 3:4], rotation=a[:, 4:8], scale=a[:, 8:11], xyz=a[:, 11:14])
