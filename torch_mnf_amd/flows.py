"""Flow modules with the reference's API, computed by libmnf_hip.so on MI355X.

Drop-in boundary: each class keeps the reference's constructor
signature, attribute names and ``state_dict`` keys, and the duck-typed Flow interface

    flow.forward(z) -> (x, log_det)        flow.inverse(x) -> (z, log_det)

so they slot into ``NormalizingFlow`` / ``NormalizingFlowModel`` (here or the reference's)
and into MNF layers unchanged.  The arithmetic is a C-ABI call on raw device pointers and
the current HIP stream; PyTorch only owns the memory.  Inputs must be fp32 tensors on the
GPU -- there is no CPU path (the CPU restatement lives in ``oracle/`` and is test-only).

Reference classes (below /root/reference/torch_mnf):
  AffineHalfFlow  flows/affine_half_flow.py:20-66     NSF_CL   flows/spline_flow.py:238-285
  RNVP            flows/rnvp.py:7-39                  Glow     flows/glow.py:5-37
  AffineConstantFlow / ActNormFlow  flows/affine_constant_flow.py:7-50
  NormalizingFlow / NormalizingFlowModel  flows/core.py:10-55     MLP  models/mlp.py:4-12
"""
from __future__ import annotations

import contextlib
import ctypes
import math
from collections.abc import Sequence

import torch
from torch import Tensor, nn

import os

from . import _dispatch
from . import _lib
from . import dist as _dist
from ._lib import MnfHipError

# MNF_FP32_MFMA=1: run the fp32 MFMA kernels instead of the split (f16 hi + lo) ones, for A/B measurements
_FP32_MFMA_ENV = os.environ.get("MNF_FP32_MFMA", "0") == "1"
# (every shape / row-count threshold of this module lives in _dispatch.py)
# MNF_NO_RUN_FUSION=1: NormalizingFlow launches every layer separately (per-layer measurements)
_NO_RUN_FUSION_ENV = os.environ.get("MNF_NO_RUN_FUSION", "0") == "1"

__all__ = [
    "MLP", "AffineHalfFlow", "NSF_CL", "NSF_AR", "RNVP", "AffineConstantFlow", "ActNormFlow", "Glow",
    "NormalizingFlow", "NormalizingFlowModel", "StandardNormal", "FusedSplineBlock", "FusedAffineStack", "rqs",
]


class MLP(nn.Sequential):
    """Linear / LeakyReLU(0.2) chain, last activation dropped (models/mlp.py:4-12).

    Holds the conditioner's parameters under the reference's key names
    (``0.weight``, ``0.bias``, ``2.weight`` ...).  The HIP kernels read the parameters
    directly; this module's own ``forward`` is never on the hot path."""

    def __init__(self, *layer_sizes: int, leaky_a: float = 0.2) -> None:
        layers: list[nn.Module] = []
        for s1, s2 in zip(layer_sizes, layer_sizes[1:]):
            layers.append(nn.Linear(s1, s2))
            layers.append(nn.LeakyReLU(leaky_a))
        super().__init__(*layers[:-1])
        self.layer_sizes = tuple(int(s) for s in layer_sizes)


class MaskedLinear(nn.Linear):
    """A dense layer with a binary mask on its weights (layers/made.py:11-25): ``x @ (W.T * mask) + b``; ``mask`` is a
    buffer of shape (n_in, n_out).  Inside MAF / IAF the kernels read weight, bias and mask directly; this ``forward``
    is the reference's formula on device tensors for anyone who calls the network on its own."""

    def __init__(self, n_in: int, n_out: int, bias: bool = True) -> None:
        super().__init__(n_in, n_out, bias)
        self.register_buffer("mask", torch.ones(n_in, n_out))

    def set_mask(self, mask) -> None:
        self.mask = torch.as_tensor(mask).to(self.weight.device)

    def forward(self, x: Tensor) -> Tensor:
        return x @ (self.weight.T * self.mask) + self.bias


class MADE(nn.Sequential):
    """Masked autoregressive MLP (layers/made.py:28-94; Germain et al. 2015, in karpathy's construction): MaskedLinear
    layers with ReLU between them whose masks make output i (and i + n_in, ...) a function of the inputs ordered before
    input i only.  Same constructor, attributes and state_dict keys (``{2l}.weight / .bias / .mask``) as the reference;
    the unit degrees come from ``numpy.random.RandomState(seed)`` exactly as there, so the masks are the reference's."""

    def __init__(self, n_in: int, hidden_sizes, n_out: int, num_masks: int = 1, natural_ordering: bool = False) -> None:
        if n_out % n_in:
            raise AssertionError("n_out must be integer multiple of n_in")  # (made.py:39)
        self.n_in, self.n_out, self.hidden_sizes = int(n_in), int(n_out), list(hidden_sizes)
        sizes = [self.n_in, *self.hidden_sizes, self.n_out]
        layers: list[nn.Module] = []
        for a, b in zip(sizes, sizes[1:]):
            layers += [MaskedLinear(a, b), nn.ReLU()]
        super().__init__(*layers[:-1])
        self.natural_ordering, self.num_masks, self.seed = natural_ordering, num_masks, 0
        self.m: dict = {}
        self.update_masks()

    def update_masks(self) -> None:
        """(made.py:58-94) degrees of the inputs (natural order or a permutation), of every hidden unit (uniform between
        the previous layer's smallest degree and n_in - 2), connection rules <= between hidden layers and < into the
        outputs; the next call uses the next seed when ``num_masks > 1``."""
        import numpy as np

        if self.m and self.num_masks == 1:
            return
        rng = np.random.RandomState(self.seed)
        self.seed = (self.seed + 1) % self.num_masks
        self.m[-1] = np.arange(self.n_in) if self.natural_ordering else rng.permutation(self.n_in)
        for layer, size in enumerate(self.hidden_sizes):
            self.m[layer] = rng.randint(self.m[layer - 1].min(), self.n_in - 1, size=size)
        n = len(self.hidden_sizes)
        masks = [self.m[layer - 1][:, None] <= self.m[layer][None, :] for layer in range(n)]
        masks.append(self.m[n - 1][:, None] < self.m[-1][None, :])
        if self.n_out > self.n_in:
            masks[-1] = np.concatenate([masks[-1]] * (self.n_out // self.n_in), axis=1)
        for layer, mask in zip([m for m in self if isinstance(m, MaskedLinear)], masks):
            layer.set_mask(mask)


def _dist_world() -> int:
    """Ranks of the default ``torch.distributed`` group (1: not initialised)."""
    import torch.distributed as td

    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def _require_mlp(*nets: nn.Module) -> None:
    """The kernels evaluate the conditioner themselves: Linear / LeakyReLU(0.2) chains (the reference's MLP with its
    default slope, models/mlp.py:4-12).  Any other conditioner class or slope cannot run on them -- say so instead of
    packing a parameter buffer of the wrong shape."""
    for net in nets:
        acts = [m for m in net.modules() if isinstance(m, nn.LeakyReLU)] if isinstance(net, nn.Module) else []
        if not isinstance(net, MLP) or any(abs(a.negative_slope - 0.2) > 1e-12 for a in acts):
            raise NotImplementedError(
                "torch_mnf_amd kernels implement the reference's MLP conditioner with LeakyReLU(0.2) only; "
                f"got {type(net).__name__}" + (f" with slopes {[a.negative_slope for a in acts]}" if acts else ""))


# The split (f16 hi + lo) kernels carry ~22 bits per product (csrc/mnf_split.h).  In sums over 16-32 terms that
# averages out to 1-2e-7; through a hidden layer of one or two units a sum IS one or two products, and a few exp(s)
# later the result can sit 1e-5 from float64 (seeded fuzz, round 1: 1.2e-5 with a one-unit layer).  Conditioners with a
# hidden layer narrower than this run on the fp32 MFMA kernels instead.
_MIN_SPLIT_HIDDEN = 4


# MNF_CHECK_PARAMS=N (N >= 1): on every N-th use of a cached operand image, compare the parameters it was packed from
# with the parameters as they are now (exact comparison on the device, one host round trip) and raise if they differ.
# The caches are keyed on (data_ptr, _version) of every parameter; a write THROUGH ``p.data`` (``p.data.mul_(2)``,
# ``p.data.clamp_()``, ``p.data.copy_()``: weight clipping, EMA, some init code) bumps neither, and the kernels would
# go on computing with the old weights, silently.  ``invalidate()`` is the remedy; this switch is the detector.
_CHECK_PARAMS_EVERY = int(os.environ.get("MNF_CHECK_PARAMS", "0") or 0)
_check_params_calls = 0


def _check_params_fresh(params, packed_flat: Tensor | None, what: str) -> None:
    """Raise if ``packed_flat`` (what the operand images were packed from) no longer equals the parameters."""
    global _check_params_calls
    if _CHECK_PARAMS_EVERY <= 0 or packed_flat is None:
        return
    _check_params_calls += 1
    if _check_params_calls % _CHECK_PARAMS_EVERY:
        return
    if packed_flat.is_cuda and torch.cuda.is_current_stream_capturing():
        return  # (the comparison reads a result back: not inside a hipGraph capture)
    with torch.no_grad():
        now = torch.cat([p.detach().reshape(-1) for p in params]).to(packed_flat.device, torch.float32)
        # (bitwise: NaN parameters compare equal to themselves, -0.0 differs from +0.0)
        same = now.numel() == packed_flat.numel() and bool(torch.equal(now.view(torch.int32), packed_flat.view(torch.int32)))
    if not same:
        raise RuntimeError(
            f"torch_mnf_amd: the parameters of {what} changed without their version counters moving (a write "
            "through p.data, or memory rewritten behind PyTorch's back): the packed operand images are stale. "
            "Call .invalidate() on the layer (or on the NormalizingFlow / MNFLinear) after such a write, or write "
            "with `with torch.no_grad(): p.mul_(2)` instead of `p.data.mul_(2)`.")


def _narrow_hidden(h_sizes) -> bool:
    return len(h_sizes) > 0 and min(h_sizes) < _MIN_SPLIT_HIDDEN


def _flat_gen(module) -> int:
    """Generation counter of the train.FlatParameters buffer the module's parameters live in (0: none).  Optimizers
    that rewrite that buffer through its raw pointer bump it; every packed-parameter cache key includes it, because
    such a write does not move a parameter's own version counter."""
    flat = module.__dict__.get("_mnf_flat")
    return 0 if flat is None else flat.generation


def _flat_home_of(module: nn.Module, params: list) -> tuple | None:
    """(FlatParameters, offset, length) when ``params`` (the module's packed parameters, in order) are back-to-back
    views of one train.FlatParameters buffer AND their ``.grad`` are still that object's gradient views -- the kernels
    then read the slice (no concatenation) and ADD their gradients to the gradient slice in place (no per-parameter
    ``add_``) -- else None.  The slice lookup is cached per FlatParameters object; what can change behind its back is
    re-validated on every call (see _AffineRun.flat_home for the cases)."""
    flat = module.__dict__.get("_mnf_flat")
    if flat is None or not params:
        return None
    cache = module.__dict__.setdefault("_home_cache", {})
    key = (id(params[0]), id(params[-1]), len(params))  # (a module may ask for several parameter lists)
    cached = cache.get(key)
    if cached is None or cached[0] is not flat or cached[1] is not params[0]:
        sl = flat.slice_of(params)
        i0 = next((i for i, q in enumerate(flat.params) if q is params[0]), None) if sl is not None else None
        cached = cache[key] = (flat, params[0], sl, i0)
    _, _, sl, i0 = cached
    if sl is None or i0 is None or not flat.home_is_valid(params):
        return None
    views = flat._grad_views
    for k, prm in enumerate(params):  # every parameter's gradient is still its view of the flat gradient buffer
        if prm.grad is not views[i0 + k]:
            return None
    return flat, sl[0], sl[1]


def _home_stand_in(module: nn.Module, device) -> Tensor:
    """A one-element tensor that carries ``requires_grad`` through an autograd function whose parameters live in a
    flat home (the function reads the parameter slice and adds to the gradient slice itself)."""
    t = module.__dict__.get("_stand_in_t")
    if t is None or t.device != device:
        t = module.__dict__["_stand_in_t"] = torch.zeros(1, device=device, requires_grad=True)
    return t


def _ptr(t: Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _device_input(t: Tensor, what: str) -> Tensor:
    if not isinstance(t, Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"torch_mnf_amd: {what} must be a GPU tensor (got {getattr(t, 'device', type(t))}); "
            "the HIP path has no CPU fallback"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"torch_mnf_amd: {what} must be float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"torch_mnf_amd: {what} must be (rows, dim), got {tuple(t.shape)}")
    return t.detach().contiguous()


def _grad_input(t: Tensor) -> Tensor:
    """Same checks as _device_input but keeps the autograd link (training path)."""
    _device_input(t, "input")
    return t.contiguous()


def _empty_result(x: Tensor, accum: Tensor | None):
    """rows == 0: nothing to launch (an empty tensor has no device pointer to hand over)."""
    return torch.empty_like(x), (None if accum is not None else x.new_empty(0))


def _wants_grad(module: nn.Module, x: Tensor) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


def _grad_scale(gy: Tensor | None, gl: Tensor | None, rows: int, dim: int, device) -> Tensor:
    """Device float: the power of two that brings the incoming gradients near 1 (the split gradient kernel carries
    them as f16 pairs; a mean loss makes them ~1/rows).  Taken from a sample of the rows, without a host round trip."""
    if gy is None and gl is None:
        return torch.ones(1, dtype=torch.float32, device=device)
    scale = torch.empty(1, dtype=torch.float32, device=device)
    _lib.check("mnf_affine_half_grad_scale", _lib.load().mnf_affine_half_grad_scale(
        _ptr(gy), _ptr(gl), rows, dim, scale.data_ptr(), _stream()))
    return scale


bwd_kernel_events: list | None = None  # set to a list to collect (start, end) events of every split gradient launch
rnvp_bwd_kernel_events: list | None = None  # likewise for launch B-ts of every RNVP gradient pass


def _bwd_split_workspace(lib, f, rows: int, device) -> Tensor | None:
    """Device floats for the split gradient kernel's two-stage flush (one buffer serves every layer of a run)."""
    n = lib.mnf_affine_half_bwd_split_workspace(rows, f.dim, len(f.h_sizes), f._hid)
    return torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None


def _ahf_layer_backward(lib, f, x_in: Tensor, gy, gl, gx: Tensor, grad_flat_ptr, flat_ptr, bwd_image_ptr, scale, cold,
                        inverse: bool, work: Tensor | None = None, lp_scratch: Tensor | None = None,
                        y_out: Tensor | None = None) -> bool:
    """Gradients of ONE AffineHalfFlow layer: the split-MFMA kernel (+ its fp32 fix-up pass over the tiles it handed
    back), else the fp32-MFMA kernel, else the generic one.  grad_x is written, the flat gradient added to.

    ``lp_scratch`` (rows x dim, uninitialised) asks for the log-prob form of the split kernel: ``gl`` is d loss / d
    log p per row, ``gy`` is None -- the kernel forms grad_y = -y gl itself (mnf_affine_half_bwd_split_lp).  Returns
    False when that form does not exist for this call (nothing has been computed: the caller materialises grad_y)."""
    rows, hid = x_in.shape[0], (len(f.h_sizes), f._hid)
    index = f._bwd_index(x_in.device) if grad_flat_ptr is not None and not f.force_generic else None
    rc = _lib.MNF_ERR_UNSUPPORTED
    if lp_scratch is not None:
        if index is None or bwd_image_ptr is None or gl is None:
            return False
        cap = cold.numel() - 1
        marks = None
        if bwd_kernel_events is not None:
            marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            marks[0].record()
        rc = lib.mnf_affine_half_bwd_split_lp(
            x_in.data_ptr(), gl.data_ptr(), lp_scratch.data_ptr(), gx.data_ptr(), grad_flat_ptr, bwd_image_ptr,
            index.data_ptr(), rows, f.dim, int(bool(f.parity)), int(inverse), *hid, scale.data_ptr(), cold.data_ptr(),
            cap, _ptr(work), 0 if work is None else work.numel(), _stream())
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            return False
        if marks is not None:
            marks[1].record()
            bwd_kernel_events.append(marks)
        _lib.check("mnf_affine_half_bwd_split_lp", rc)
        _lib.check("mnf_affine_half_bwd_mfma_tiles", lib.mnf_affine_half_bwd_mfma_tiles(
            x_in.data_ptr(), lp_scratch.data_ptr(), gl.data_ptr(), gx.data_ptr(), grad_flat_ptr, flat_ptr,
            index.data_ptr(), rows, f.dim, int(bool(f.parity)), int(inverse), *hid, cold.data_ptr(), cap, _stream()))
        return True
    if index is not None and bwd_image_ptr is not None:
        cap = cold.numel() - 1
        marks = None
        if bwd_kernel_events is not None:  # bench.py: (start, end) HIP events around the gradient kernel proper
            marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            marks[0].record()
        rc = lib.mnf_affine_half_bwd_split(
            x_in.data_ptr(), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, bwd_image_ptr, index.data_ptr(), rows,
            f.dim, int(bool(f.parity)), int(inverse), *hid, scale.data_ptr(), cold.data_ptr(), cap, _ptr(work),
            0 if work is None else work.numel(), _stream())
        if marks is not None:
            marks[1].record()
            bwd_kernel_events.append(marks)
        if rc == _lib.MNF_OK:
            rc = lib.mnf_affine_half_bwd_mfma_tiles(
                x_in.data_ptr(), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, flat_ptr, index.data_ptr(), rows,
                f.dim, int(bool(f.parity)), int(inverse), *hid, cold.data_ptr(), cap, _stream())
    if rc == _lib.MNF_ERR_UNSUPPORTED and index is not None:
        n_ws = lib.mnf_affine_half_bwd_mfma_workspace(rows, f.dim, *hid) if _lib.deterministic() else 0
        if n_ws > 0:  # MNF_DETERMINISTIC=1: the workgroups' sums as blocks, added in a fixed order
            ws = torch.empty(n_ws, dtype=torch.float32, device=x_in.device)
            rc = lib.mnf_affine_half_bwd_mfma_det(
                x_in.data_ptr(), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, flat_ptr, index.data_ptr(), rows,
                f.dim, int(bool(f.parity)), int(inverse), *hid, ws.data_ptr(), n_ws, _stream())
        else:
            rc = lib.mnf_affine_half_bwd_mfma(
                x_in.data_ptr(), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, flat_ptr, index.data_ptr(), rows,
                f.dim, int(bool(f.parity)), int(inverse), *hid, _stream())
    if rc == _lib.MNF_ERR_UNSUPPORTED and flat_ptr is not None and f.force_generic != 1 \
            and (f.force_generic == 2 or (rows >= _dispatch.RT_MIN_ROWS and not f._fp32_request())) and (y_out is not None or not inverse or not f.scale):
        # no per-shape gradient kernel: the run-time-shaped matrix-core one (any 1..4 hidden layers of widths 4..64)
        sc = scale if scale is not None else _grad_scale(gy, gl, rows, f.dim, x_in.device)
        rc = lib.mnf_affine_half_bwd_rt(
            x_in.data_ptr(), _ptr(y_out), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, flat_ptr, sc.data_ptr(), rows,
            f.dim, int(bool(f.parity)), int(inverse), *hid, int(f.scale), int(f.shift), _stream())
    if rc == _lib.MNF_ERR_UNSUPPORTED:  # no matrix-core gradient kernel for this shape at all
        rc = lib.mnf_affine_half_bwd(
            x_in.data_ptr(), _ptr(gy), _ptr(gl), gx.data_ptr(), grad_flat_ptr, flat_ptr, rows, f.dim,
            int(bool(f.parity)), int(inverse), *hid, int(f.scale), int(f.shift), _stream())
    _lib.check("mnf_affine_half_bwd", rc)
    if not f.force_generic:
        _lib.note_generic("AffineHalfFlow.backward", rows, f"dim={f.dim}, hidden={f.h_sizes}")
    return True


class _AffineHalfFn(torch.autograd.Function):
    """AffineHalfFlow with gradients: forward = the usual kernel, backward = mnf_affine_half_bwd
    (recomputes the conditioner, returns grad wrt x and wrt the flat parameter vector; autograd's
    own cat-backward then scatters that vector onto s_net / t_net parameters)."""

    @staticmethod
    def forward(ctx, x, flat_with_grad, module, inverse):
        flat, image = module._packed(x.device)
        y = torch.empty_like(x)
        ld = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_affine_half", _lib.load().mnf_affine_half(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), 0, _ptr(flat), _ptr(image), _ptr(module._split_image(x.device)),
            x.shape[0], module.dim,
            int(bool(module.parity)), int(inverse), len(module.h_sizes), module._hid, int(module.scale),
            int(module.shift), module._force_code(image), _stream()))
        ctx.module, ctx.inverse = module, inverse
        # (y too: the run-time-shaped gradient kernel forms the inverse direction's g_s = -grad_y y - grad_ld from it; the
        #  next layer keeps its input alive anyway)
        ctx.save_for_backward(x, flat if flat is not None else x.new_empty(0), y)
        return y, ld

    @staticmethod
    def backward(ctx, grad_y, grad_ld):
        x, flat, y_out = ctx.saved_tensors
        m = ctx.module
        gy = None if grad_y is None else grad_y.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        grad_x = torch.empty_like(x)
        grad_flat = torch.zeros_like(flat)
        has = flat.numel() > 0
        bwd = (m._bwd_split_image(x.device, flat)
               if has and not m.force_generic and x.shape[0] >= _dispatch.BWD_SPLIT_MIN_ROWS else None)
        scale = cold = work = None
        if bwd is not None:
            scale = _grad_scale(gy, gl, x.shape[0], m.dim, x.device)
            cold = torch.zeros((x.shape[0] + 15) // 16 + 1, dtype=torch.int32, device=x.device)
            work = _bwd_split_workspace(_lib.load(), m, x.shape[0], x.device)
        _ahf_layer_backward(_lib.load(), m, x, gy, gl, grad_x, _ptr(grad_flat) if has else None,
                            _ptr(flat) if has else None, _ptr(bwd), scale, cold, ctx.inverse, work, y_out=y_out)
        return grad_x, (grad_flat if flat.numel() else None), None, None


class _NsfFn(torch.autograd.Function):
    """NSF_CL with gradients (mnf_nsf_cl_bwd: recompute both half-steps, reverse-mode through the
    spline and the conditioner nets)."""

    @staticmethod
    def forward(ctx, x, flat_with_grad, module, inverse):
        flat, image = module._packed(x.device)
        y = torch.empty_like(x)
        ld = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_nsf_cl", _lib.load().mnf_nsf_cl(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), 0, _ptr(flat), _ptr(image), _ptr(module._split_image(x.device)),
            x.shape[0], module.dim,
            module.K, float(module.B), int(inverse), len(module.h_sizes), module._hid,
            module._force_code(image), _stream()))
        if not module.force_generic and not module._pad_half():  # (a padded-twin shape is here by choice: few rows)
            _lib.note_generic("NSF_CL", x.shape[0], f"dim={module.dim}, K={module.K}, hidden={module.h_sizes}")
        ctx.module, ctx.inverse = module, inverse
        # (y too: the tile gradient kernel reads the second net's conditioner input out of it instead of recomputing the
        #  first half-step; the next layer keeps its input alive anyway)
        ctx.save_for_backward(x, flat, y)
        return y, ld

    @staticmethod
    def backward(ctx, grad_y, grad_ld):
        x, flat, y_out = ctx.saved_tensors
        m = ctx.module
        gy = None if grad_y is None else grad_y.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        grad_x = torch.empty_like(x)
        grad_flat = torch.zeros_like(flat)
        lib = _lib.load()
        rows = x.shape[0]
        args = (rows, m.dim, m.K, float(m.B), int(ctx.inverse), len(m.h_sizes), m._hid)
        # the tile kernel (conditioner on the matrix cores, 16 rows per wave) where it exists -- dim a multiple of 8 up to
        # 64, hidden width <= 16, K in {5, 8} (10 up to dim 32) --, followed by its fp32 fix-up pass over the tiles it handed back; else
        # the generic kernel
        table = None
        if not (m.force_generic or m.force_fp32_mfma or _FP32_MFMA_ENV or _dispatch.NSF_BWD_KERNEL == "generic"
                or rows * m.dim >= (1 << 31)):
            table = m._bwd_tile_tables(x.device)
        marks = None
        if bwd_kernel_events is not None and table:  # bench.py --workload c3t
            marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if table:
            idx, flush, n_split, n_plain = table
            image = torch.empty(n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32, device=x.device)
            _lib.check("mnf_pack_gather_split", lib.mnf_pack_gather_split(
                flat.data_ptr(), idx.data_ptr(), image.data_ptr(), n_split, n_plain, _stream()))
            scale = _grad_scale(gy, gl, rows, m.dim, x.device)
            cap = (rows + 15) // 16
            cold = torch.zeros(2 * cap + 2, dtype=torch.int32, device=x.device)
            n_work = lib.mnf_nsf_cl_bwd_tile_workspace(rows, m.dim, m.K, len(m.h_sizes), m._hid)
            work = torch.empty(n_work, dtype=torch.float32, device=x.device)
            if marks is not None:
                marks[0].record()
            rc = lib.mnf_nsf_cl_bwd_tile(
                x.data_ptr(), y_out.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(),
                image.data_ptr(), flush.data_ptr(), *args, scale.data_ptr(), cold.data_ptr(), cap, work.data_ptr(), n_work, _stream())
            if rc != _lib.MNF_ERR_UNSUPPORTED:  # (unsupported: e.g. a view at an odd storage offset -- the kernels below take it)
                _lib.check("mnf_nsf_cl_bwd_tile", rc)
                if marks is not None:
                    marks[1].record()
                    bwd_kernel_events.append(marks)
                _lib.check("mnf_nsf_cl_bwd_tile_fixup", lib.mnf_nsf_cl_bwd_tile_fixup(
                    x.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(), *args,
                    cold.data_ptr(), cap, _stream()))
                return grad_x, grad_flat, None, None
        if m.force_generic != 1 and _dispatch.NSF_BWD_KERNEL != "generic" and (
                m.force_generic == 2 or (rows >= _dispatch.RT_MIN_ROWS and not m._fp32_request())):
            # no per-shape gradient kernel: the run-time-shaped matrix-core one (any dim, K <= 16, hidden widths 4..64)
            scale = _grad_scale(gy, gl, rows, m.dim, x.device)
            rc = lib.mnf_nsf_cl_bwd_rt(
                x.data_ptr(), y_out.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(),
                scale.data_ptr(), *args, _stream())
            if rc != _lib.MNF_ERR_UNSUPPORTED:
                _lib.check("mnf_nsf_cl_bwd_rt", rc)
                return grad_x, grad_flat, None, None
        _lib.check("mnf_nsf_cl_bwd", lib.mnf_nsf_cl_bwd(
            x.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(), *args, _stream()))
        if not m.force_generic and _dispatch.NSF_BWD_KERNEL != "generic" and not m._pad_half():
            _lib.note_generic("NSF_CL.backward", rows, f"dim={m.dim}, K={m.K}, hidden={m.h_sizes}")
        return grad_x, grad_flat, None, None


class _RnvpFn(torch.autograd.Function):
    """RNVP with gradients; the backward pass sees the same mask (explicit or regenerated from the seed)."""

    @staticmethod
    def forward(ctx, z, flat_with_grad, module, mask, seed, home=None):
        # home = (FlatParameters, offset, length): flat_with_grad is a stand-in that only carries requires_grad; the
        # kernels read the parameter slice and backward ADDS to the gradient slice in place
        few = module._few(z.shape[0], mask is not None)
        flat, image = module._packed(z.device, images=not few)
        ctx.home = home
        x = torch.empty_like(z)
        ld = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        lib = _lib.load()
        # a large batch: the split forward kernels can keep y = net(mask * z) (64 floats per row) for the matrix-core
        # gradient pass, whose first launch then skips its own sweep over z
        y, wrote = None, ctypes.c_int(0)
        if (not few and z.shape[0] >= _dispatch.RNVP_KEEP_Y_MIN_ROWS and not module.force_generic
                and not _dispatch.RNVP_BWD_GENERIC and not _dispatch.rnvp_bwd_small(z.shape[0], module.dim)):
            per_row = lib.mnf_rnvp_y_floats_per_row(len(module.h_sizes), module._hid)
            if per_row > 0:
                y = torch.empty(z.shape[0], per_row, dtype=torch.float32, device=z.device)
        _lib.check("mnf_rnvp_seeded_train", lib.mnf_rnvp_seeded_train(
            z.data_ptr(), _ptr(mask), seed, x.data_ptr(), ld.data_ptr(), 0, _ptr(flat), _ptr(image),
            None if few else _ptr(module._split_image(z.device)), z.shape[0],
            module.dim, len(module.h_sizes), module._hid,
            int(module.force_generic) if few else module._force_code(image), _ptr(y), ctypes.byref(wrote),
            _stream()))
        ctx.module, ctx.seed, ctx.mask = module, seed, mask
        ctx.kept_y = y if wrote.value else None
        ctx.save_for_backward(z, flat)
        return x, ld

    @staticmethod
    def backward(ctx, grad_x, grad_ld):
        z, flat = ctx.saved_tensors
        m = ctx.module
        gx = None if grad_x is None else grad_x.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        grad_z = torch.empty_like(z)
        home = ctx.home
        lib = _lib.load()
        if gx is None and gl is None:
            return grad_z.zero_(), (None if home is not None else torch.zeros_like(flat)), None, None, None, None
        if home is not None:
            grad_flat, ret_flat = home[0].grad[home[1]:home[1] + home[2]], None
        else:
            grad_flat = ret_flat = torch.zeros_like(flat)
        # the matrix-core gradient kernels (two launches + the fp32 fix-up over flagged row groups); shapes they do not
        # cover, the fp32 switches and force_generic take the generic kernel
        # (few rows or a narrow layer: the one-launch generic kernel is the faster one -- the matrix-core pass is four
        #  launches, the first a single workgroup's sweep over all dims: tools/time_rnvp_small.py)
        # a batch of a few hundred rows (the reference trains at 128): the latency kernel as a grid, a workgroup per
        # two rows writing its own copy of the parameter gradients, then one reduction launch
        if not (m.force_generic or _dispatch.RNVP_BWD_GENERIC or _dispatch.RNVP_BWD_FEW_GRID_OFF):
            n_ws = lib.mnf_rnvp_bwd_few_workspace_floats(z.shape[0], m.dim, len(m.h_sizes), m._hid)
            if n_ws > 0:
                ws = torch.empty(n_ws, dtype=torch.float32, device=z.device)
                _lib.check("mnf_rnvp_bwd_few", lib.mnf_rnvp_bwd_few(
                    z.data_ptr(), _ptr(ctx.mask), ctx.seed, _ptr(gx), _ptr(gl), grad_z.data_ptr(), grad_flat.data_ptr(),
                    flat.data_ptr(), ws.data_ptr(), z.shape[0], m.dim, len(m.h_sizes), m._hid, _stream()))
                return grad_z, ret_flat, None, None, None, None
        small = _dispatch.rnvp_bwd_small(z.shape[0], m.dim)
        bwd = None if (m.force_generic or _dispatch.RNVP_BWD_GENERIC or small) else m._bwd_image(z.device, flat)
        split = m._split_image(z.device) if bwd is not None else None
        if bwd is not None and split is not None:
            rows = z.shape[0]
            work = m._bwd_workspace(lib, rows, z.device)
            scale = _grad_scale(gx, gl, rows, m.dim, z.device)
            def go(phases):
                return lib.mnf_rnvp_bwd_mfma_phases(
                    z.data_ptr(), _ptr(ctx.mask), ctx.seed, _ptr(gx), _ptr(gl), grad_z.data_ptr(), grad_flat.data_ptr(),
                    flat.data_ptr(), split.data_ptr(), bwd.data_ptr(), scale.data_ptr(), work.data_ptr(), work.numel(),
                    rows, m.dim, len(m.h_sizes), m._hid, phases, _ptr(ctx.kept_y), _stream())

            if rnvp_bwd_kernel_events is None:
                rc = go(15)
            else:  # bench.py: HIP events around launch B-ts, the pass's dominant kernel
                rc = go(1)
                if rc == _lib.MNF_OK:
                    marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    marks[0].record()
                    rc = go(2)
                    marks[1].record()
                    rnvp_bwd_kernel_events.append(marks)
                if rc == _lib.MNF_OK:
                    rc = go(12)
            if rc != _lib.MNF_ERR_UNSUPPORTED:
                _lib.check("mnf_rnvp_bwd_mfma", rc)
                return grad_z, ret_flat, None, None, None, None
        if m.force_generic != 1 and not _dispatch.RNVP_BWD_GENERIC and (
                m.force_generic == 2 or (z.shape[0] >= _dispatch.RT_MIN_ROWS and not m._fp32_request())):
            # no per-shape gradient kernel: the run-time-shaped matrix-core one (1..4 conditioner layers of widths 4..128)
            scale = _grad_scale(gx, gl, z.shape[0], m.dim, z.device)
            rc = lib.mnf_rnvp_bwd_rt(
                z.data_ptr(), _ptr(ctx.mask), ctx.seed, _ptr(gx), _ptr(gl), grad_z.data_ptr(), grad_flat.data_ptr(),
                flat.data_ptr(), scale.data_ptr(), z.shape[0], m.dim, len(m.h_sizes), m._hid, _stream())
            if rc != _lib.MNF_ERR_UNSUPPORTED:
                _lib.check("mnf_rnvp_bwd_rt", rc)
                return grad_z, ret_flat, None, None, None, None
        _lib.check("mnf_rnvp_bwd", lib.mnf_rnvp_bwd(
            z.data_ptr(), _ptr(ctx.mask), ctx.seed, _ptr(gx), _ptr(gl), grad_z.data_ptr(), grad_flat.data_ptr(),
            flat.data_ptr(), z.shape[0], m.dim, len(m.h_sizes), m._hid, _stream()))
        return grad_z, ret_flat, None, None, None, None


class _AffineConstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, t, inverse):
        y = torch.empty_like(x)
        _lib.check("mnf_affine_const", _lib.load().mnf_affine_const(
            x.data_ptr(), y.data_ptr(), s.contiguous().data_ptr(), t.contiguous().data_ptr(), None, 0, None,
            x.shape[0], x.shape[1], int(inverse), _stream()))
        ctx.inverse = inverse
        ctx.save_for_backward(x, y, s)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, s = ctx.saved_tensors
        gy = grad_y.contiguous()
        gx = torch.empty_like(x)
        gs, gt = torch.zeros_like(s), torch.zeros_like(s)
        _lib.check("mnf_affine_const_bwd", _lib.load().mnf_affine_const_bwd(
            x.data_ptr(), y.data_ptr(), gy.data_ptr(), s.contiguous().data_ptr(), gx.data_ptr(), gs.data_ptr(),
            gt.data_ptr(), x.shape[0], x.shape[1], int(ctx.inverse), _stream()))
        return gx, gs, gt, None


_DEVICE_MASKS = False


@contextlib.contextmanager
def device_drawn_masks():
    """Inside: RNVP layers called without mask or seed draw their Bernoulli mask as a device tensor (torch.bernoulli,
    the reference's rnvp.py:28) instead of hashing a host-drawn seed in the kernel.  A step recorded in a hipGraph
    (train.GraphedStep) needs this: kernel arguments are frozen by the capture, device draws are not."""
    global _DEVICE_MASKS
    before, _DEVICE_MASKS = _DEVICE_MASKS, True
    try:
        yield
    finally:
        _DEVICE_MASKS = before


_LINEAR_ROWS_INDEX: dict = {}  # (dim, device) -> index table of the MFMA kernel's operand image (None: no such kernel)


def _linear_rows(lib, x: Tensor, W: Tensor, y: Tensor) -> None:
    """y = x @ W for a (dim, dim) W that changes every call (the training path): on the MFMA kernel where dim has one
    (the operand image is packed from W by one small gather launch), else on the generic kernel."""
    dim, key = x.shape[1], (x.shape[1], x.device)
    if key not in _LINEAR_ROWS_INDEX:
        n = lib.mnf_linear_rows_image_floats(dim)
        table = None
        if n > 0:
            idx = (ctypes.c_int32 * n)()
            _lib.check("mnf_linear_rows_image_index", lib.mnf_linear_rows_image_index(dim, idx))
            table = torch.frombuffer(idx, dtype=torch.int32).clone().to(x.device)
        _LINEAR_ROWS_INDEX[key] = table
    table = _LINEAR_ROWS_INDEX[key]
    if table is None:
        _lib.check("mnf_linear_rows", lib.mnf_linear_rows(x.data_ptr(), W.data_ptr(), y.data_ptr(), x.shape[0], dim,
                                                          _stream()))
        return
    img = _linear_rows_image(lib, W, table)
    _lib.check("mnf_linear_rows_img", lib.mnf_linear_rows_img(x.data_ptr(), img.data_ptr(), y.data_ptr(), x.shape[0],
                                                              dim, _stream()))


def _linear_rows_image(lib, W: Tensor, table: Tensor) -> Tensor:
    """W (dim, dim, contiguous) in the MFMA kernels' operand order: one small gather launch."""
    img = torch.empty(table.numel(), dtype=torch.float32, device=W.device)
    _lib.check("mnf_pack_gather", lib.mnf_pack_gather(W.data_ptr(), table.data_ptr(), img.data_ptr(), table.numel(),
                                                      _stream()))
    return img


def _linear_rows_table(lib, dim: int, device) -> Tensor | None:
    key = (dim, device)
    if key not in _LINEAR_ROWS_INDEX:
        n = lib.mnf_linear_rows_image_floats(dim)
        table = None
        if n > 0:
            idx = (ctypes.c_int32 * n)()
            _lib.check("mnf_linear_rows_image_index", lib.mnf_linear_rows_image_index(dim, idx))
            table = torch.frombuffer(idx, dtype=torch.int32).clone().to(device)
        _LINEAR_ROWS_INDEX[key] = table
    return _LINEAR_ROWS_INDEX[key]


_PAIR_FUSION_DIMS = (16, 32, 64)  # mnf_glow_actnorm_inv / _bwd (csrc/mnf_glow_actnorm.hip)
_NO_PAIR_FUSION_ENV = os.environ.get("MNF_NO_PAIR_FUSION", "0") == "1"
# MNF_DETERMINISTIC=1 (read once by the LIBRARY: _lib.deterministic() is the one source of truth): gradient sums through
# fixed-order two-stage reductions instead of float atomics where a kernel has both (the reference's loop repeats bit for
# bit under torch.manual_seed(0), tests/test_flows.py:11).  The per-shape AffineHalfFlow and NSF_CL gradient kernels
# reduce in a fixed order in every mode; the switch adds the RNVP, MNFLinear, sample_z and [Glow, ActNorm] launches, and
# keeps shapes without a per-shape kernel off the run-time-shaped gradient kernels (atomic flushes).  What still adds
# atomically then: rows that take an fp32 fix-up pass (operands beyond the split range) and the VALU any-shape gradient
# kernels.


def _pair_bwd_workspace(rows: int, dim: int, device):
    """Block sums of the pair's gradient launch (deterministic mode), else None."""
    if not _lib.deterministic():
        return None
    n = _lib.load().mnf_glow_actnorm_inv_bwd_workspace(rows, dim)
    return torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None


class _GlowActNormInvFn(torch.autograd.Function):
    """Glow.inverse then ActNormFlow.inverse -- z = (u @ M - t) e^-s, M = W^-1 -- as ONE autograd node with one launch
    each way (glow.py:33-37, affine_constant_flow.py:22-26).  The intermediate u @ M is never written; the gradient
    launch reads u and grad_z once and produces grad_u, grad_M, grad_s and grad_t.  The pair's log|det J| (Glow's, handed
    in, minus sum s) is a second output of the same launch."""

    @staticmethod
    def forward(ctx, u, M, s, t, ld_glow):
        Mc = M.detach().contiguous()
        sc = s.detach().to(u.device, torch.float32).contiguous()
        tc = t.detach().to(u.device, torch.float32).contiguous()
        z = torch.empty_like(u)
        ld = torch.empty(1, dtype=torch.float32, device=u.device)  # Glow's log|det| - sum s, formed by the launch
        _lib.check("mnf_glow_actnorm_inv", _lib.load().mnf_glow_actnorm_inv(
            u.data_ptr(), Mc.data_ptr(), sc.data_ptr(), tc.data_ptr(), z.data_ptr(), ld_glow.detach().data_ptr(),
            ld.data_ptr(), u.shape[0], u.shape[1], _stream()))
        ctx.save_for_backward(u, Mc, sc, tc)
        ctx.ld_shape = ld_glow.shape
        ctx.set_materialize_grads(False)
        return z, ld

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_z, grad_ld):
        u, Mc, sc, tc = ctx.saved_tensors
        dim = u.shape[1]
        gz = torch.zeros_like(u) if grad_z is None else grad_z.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        gu = torch.empty_like(u)
        sums = torch.zeros(dim * dim + 2 * dim, dtype=torch.float32, device=u.device)  # grad_M | grad_s | grad_t
        gM, gs, gt = sums[:dim * dim], sums[dim * dim:dim * dim + dim], sums[dim * dim + dim:]
        work = _pair_bwd_workspace(u.shape[0], dim, u.device)
        _lib.check("mnf_glow_actnorm_inv_bwd", _lib.load().mnf_glow_actnorm_inv_bwd_det(
            u.data_ptr(), gz.data_ptr(), Mc.data_ptr(), sc.data_ptr(), tc.data_ptr(), gu.data_ptr(), gM.data_ptr(),
            gs.data_ptr(), gt.data_ptr(), _ptr(gl), u.shape[0], dim, _ptr(work), 0 if work is None else work.numel(),
            _stream()))
        return (gu if ctx.needs_input_grad[0] else None, gM.view(dim, dim), gs.view(sc.shape), gt.view(tc.shape),
                None if gl is None else gl.reshape(ctx.ld_shape))


class _LinearRowsFn(torch.autograd.Function):
    """y = x @ W with both gradients from the HIP library."""

    @staticmethod
    def forward(ctx, x, W):
        W = W.contiguous()
        y = torch.empty_like(x)
        _linear_rows(_lib.load(), x, W, y)
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, W = ctx.saved_tensors
        gy = grad_y.contiguous()
        gx = torch.empty_like(x)
        Wt = W.t().contiguous()
        lib = _lib.load()
        _linear_rows(lib, gy, Wt, gx)
        gW = torch.zeros_like(W)
        _lib.check("mnf_linear_rows_bwd_weight", lib.mnf_linear_rows_bwd_weight(
            x.data_ptr(), gy.data_ptr(), gW.data_ptr(), x.shape[0], x.shape[1], _stream()))
        return gx, gW


class _AffineRunFn(torch.autograd.Function):
    """A run of equal AffineHalfFlow layers as ONE autograd node: forward = the stack kernel (every
    intermediate written once), backward = the fp32-MFMA gradient kernel layer by layer on the saved
    intermediates.  One parameter cat / one grad scatter for the whole run instead of one per layer."""

    @staticmethod
    def forward(ctx, x, flat_with_grad, run, inverse, with_lp=False):
        """``with_lp``: the run is a whole density pass under a standard-normal base -- the stack kernel's epilogue also
        writes log p = log_det + log N(z; 0, I), the ONLY output of the node then, and the backward pass forms the last
        layer's cotangents from d loss / d log p inside the gradient kernel (mnf_affine_half_bwd_split_lp)."""
        n = len(run.layers)
        ld = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        # flat-homed parameters (train.FlatParameters): flat_with_grad is a one-element stand-in that only carries
        # requires_grad; the kernels read the parameter slice and ADD their gradients to the gradient slice in place
        home = run._home_now  # (the validated flat home launch_grad decided on for this call)
        ctx.home = home
        flat = home[0].data[home[1]:home[1] + home[2]] if home is not None else flat_with_grad.detach()
        imgs = run.images(x.device, flat)  # after a weight update: repacked from this one concatenation
        lp = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if with_lp else None
        outs = (run.launch(x, inverse, ld, False, None, keep=True, images=imgs, logprob=(lp, None) if with_lp else None)
                if imgs[0] is not None else None)
        if outs is None or (with_lp and not run.logprob_fused):
            raise MnfHipError("mnf_affine_half_stack", _lib.MNF_ERR_UNSUPPORTED, "no stack kernel for this shape")
        ctx.run, ctx.inverse, ctx.with_lp = run, inverse, with_lp
        ctx.set_materialize_grads(False)
        ctx.n = n
        if with_lp:
            ctx.save_for_backward(x, flat, *outs)  # (the last output too: the fallback's grad_y = -z g)
            return lp
        ctx.save_for_backward(x, flat, *outs[:-1])
        return (*outs, ld)

    @staticmethod
    def backward(ctx, *grads):
        run, inverse, n = ctx.run, ctx.inverse, ctx.n
        x, flat, *mids = ctx.saved_tensors
        z_last = None
        if ctx.with_lp:  # one incoming cotangent: d loss / d log p
            z_last, mids = mids[-1], mids[:-1]
            if grads[0] is None:
                return None, None, None, None, None
            grads = (None,) * n + (grads[0],)  # log p = log_det + base(z): d / d log_det = g; d / d z below
        inputs = [x] + list(mids)                     # input of the li-th applied layer
        order = list(reversed(run.layers)) if inverse else list(run.layers)
        grad_ld = grads[-1]
        gl = None if grad_ld is None else grad_ld.contiguous()
        home = ctx.home
        grad_flat = home[0].grad[home[1]:home[1] + home[2]] if home is not None else torch.zeros_like(flat)
        # offset of every layer's parameters inside the run's flat vector (model order)
        sizes = [sum(p.numel() for p in f._packed_params()) for f in run.layers]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        lib = _lib.load()
        g = grads[n - 1]
        # split gradient kernel: every layer's backward image from one launch, one gradient scale for the run (the
        # magnitude changes by e^s per layer: far inside the split range), one fix-up list per layer
        bwd = run.bwd_images(x.device, flat) if x.shape[0] >= _dispatch.BWD_SPLIT_MIN_ROWS else None
        scale = cold = work = None
        if bwd is not None:
            first = next((t for t in reversed(grads[:n]) if t is not None), None)
            scale = _grad_scale(None if first is None else first.contiguous(), gl, x.shape[0], x.shape[1], x.device)
            cold = torch.zeros((n, (x.shape[0] + 15) // 16 + 1), dtype=torch.int32, device=x.device)
            work = _bwd_split_workspace(lib, run.layers[0], x.shape[0], x.device)
        for li in range(n - 1, -1, -1):
            f = order[li]
            k = run.layers.index(f)
            gx = torch.empty_like(x)
            args = (grad_flat.data_ptr() + 4 * offs[k], flat.data_ptr() + 4 * offs[k],
                    None if bwd is None else bwd[0].data_ptr() + 4 * bwd[1] * k, scale,
                    None if cold is None else cold[k], inverse, work)
            done = False
            if ctx.with_lp and li == n - 1:  # the last layer: grad_y = -z g formed inside the gradient kernel
                if bwd is not None:
                    done = _ahf_layer_backward(lib, f, inputs[li], None, gl, gx, *args, lp_scratch=torch.empty_like(x))
                if not done:
                    g = z_last * (-gl).unsqueeze(1)
            if not done:
                gy = None if g is None else g.contiguous()
                _ahf_layer_backward(lib, f, inputs[li], gy, gl, gx, *args)
            g = gx if li == 0 or grads[li - 1] is None else gx + grads[li - 1]
        return g, (None if home is not None else grad_flat), None, None, None


class _HipFlow(nn.Module):
    """Shared plumbing: packed-parameter caches keyed on parameter versions."""

    def __init__(self) -> None:
        super().__init__()
        self._cache_key = None
        self._flat: Tensor | None = None
        self._image: Tensor | None = None
        self._index: Tensor | None = None  # device int32 gather table, built once
        self._split: Tensor | None = None  # split (f16 hi + lo) operand image, see csrc/mnf_split.h
        self._split_index = None           # (device int32 table, n_split_words, n_plain_words) or False
        self.force_generic = False  # tests: run the generic kernel even if an MFMA one exists
        self.force_fp32_mfma = False  # tests / MNF_FP32_MFMA=1: fp32 MFMA kernel instead of the split one
        # False: this conditioner always takes the fp32 MFMA kernels (set by layers whose hidden layers are so narrow
        # that a product sum has too few terms to average the split format's 2^-22 per product out, see _narrow_hidden)
        self._split_ok = True

    def _fp32_request(self) -> bool:
        """force_fp32_mfma / MNF_FP32_MFMA=1: no split-f16 arithmetic for this layer."""
        return bool(self.force_fp32_mfma or _FP32_MFMA_ENV)

    def _force_code(self, image: Tensor | None) -> int:
        """`force_generic` as the C entries take it (0: dispatch, 1: VALU kernel, 2: run-time-shaped kernel).  An fp32
        request on a shape without operand image -- no per-shape kernel, so no fp32 matrix-core one -- runs the VALU
        kernel, which is fp32 too: the run-time-shaped kernels are split arithmetic (with an image and no fp32 kernel
        for the shape the C entry does the same by itself: csrc/mnf_generic.hip)."""
        fg = int(self.force_generic)
        return 1 if not fg and image is None and self._fp32_request() else fg

    def invalidate(self) -> None:
        """Drop the packed operand images; the next call repacks them from the parameters.

        The caches are keyed on every parameter's ``(data_ptr, _version)``, which optimizers, ``load_state_dict``,
        ``copy_`` and every other in-place op on the parameter itself bump.  Writes THROUGH ``p.data``
        (``p.data.clamp_()``, ``p.data.copy_()`` -- weight clipping, EMA, some init code) do not bump the
        parameter's version counter, so after such a write call ``invalidate()`` (on the layer, or on the
        ``NormalizingFlow`` for all of its layers), or write with ``torch.no_grad(): p.clamp_()`` instead."""
        self._cache_key = None
        for attr in ("_w_key", ):
            if hasattr(self, attr):
                setattr(self, attr, None)

    # subclasses: ordered parameter list == state_dict order
    def _packed_params(self) -> list[Tensor]:
        return [p for p in self.parameters()]

    def _net_params(self, nets: Sequence[nn.Module]) -> list[Tensor]:
        """``[p for net in nets for p in net.parameters()]`` without walking the module tree on every call (a
        small-batch pass otherwise spends most of its host time in nn.Module.parameters()).  The walk is cached
        per set of nets and their direct children -- the MLP / Linear conditioners of this package; any other
        module type takes the plain walk."""
        if not all(type(n) in (MLP, nn.Linear) for n in nets):
            return [p for n in nets for p in n.parameters()]
        sig = tuple(id(c) for n in nets for c in (n, *n._modules.values()))
        cache = self.__dict__.get("_net_param_cache")
        if cache is None or cache[0] != sig:
            cache = (sig, [m for n in nets for m in n.modules()])
            self.__dict__["_net_param_cache"] = cache
        return [p for m in cache[1] for p in m._parameters.values() if p is not None]

    def _image_index_host(self):  # -> ctypes int32 array or None
        return None

    def _split_index_host(self):  # -> (ctypes int32 array, n_split_words, n_plain_words) or None
        return None

    def _split_image(self, device: torch.device) -> Tensor | None:
        """The split operand image for the current parameters, or None (no split kernel / switched off)."""
        if self.force_fp32_mfma or _FP32_MFMA_ENV or not self._split_ok:
            return None
        self._packed(device)
        return self._split

    def _packed3(self, device: torch.device):
        """(flat, image, split image) with ONE walk over the parameters (the walk is what a small-batch call
        spends its host time on)."""
        flat, image = self._packed(device)
        return flat, image, (None if (self.force_fp32_mfma or _FP32_MFMA_ENV or not self._split_ok) else self._split)

    def _device_index(self, device: torch.device) -> Tensor | None:
        """Device copy of the fp32 operand-image index table (built once), or None."""
        if self._index is None or self._index.device != device:
            host = self._image_index_host()
            self._index = None if host is None else torch.frombuffer(host, dtype=torch.int32).clone().to(device)
        return self._index

    def _device_split_index(self, device: torch.device):
        """(device index table, n_split_words, n_plain_words) of the split operand image, or False."""
        if self._split_index is None or (self._split_index and self._split_index[0].device != device):
            host = self._split_index_host()
            self._split_index = False if host is None else (
                torch.frombuffer(host[0], dtype=torch.int32).clone().to(device), host[1], host[2])
        return self._split_index

    def _packed(self, device: torch.device, images: bool = True) -> tuple[Tensor | None, Tensor | None]:
        """(flat parameters, fp32 operand image) for the current parameter values.  ``images=False``: the caller's
        kernel reads ``flat`` only (RNVP on one or two rows) -- the operand images are then not packed (two launches
        per layer per weight update saved: an MNF layer's flow_r never sees more than one row)."""
        params = self._packed_params()
        if not params:
            return None, None
        key = (device, _flat_gen(self), tuple((p.data_ptr(), p._version) for p in params))
        if key != self._cache_key:
            home = _flat_home_of(self, params) if params[0].device == device else None
            if home is not None:  # the parameters ARE one buffer: a (new) view of it, no concatenation
                self._flat = home[0].data[home[1]:home[1] + home[2]]
            else:
                flat = torch.cat([p.detach().reshape(-1) for p in params]).to(device=device, dtype=torch.float32)
                self._flat = flat.contiguous()
            self._image = self._split = None
            self.__dict__["_images_key"] = None
            self._cache_key = key
        elif _CHECK_PARAMS_EVERY:
            _check_params_fresh(params, self._flat, type(self).__name__)
        if images and self.__dict__.get("_images_key") != key:
            self._device_index(device)
            if self._index is not None:
                image = torch.empty(self._index.numel(), dtype=torch.float32, device=device)
                _lib.check("mnf_pack_gather", _lib.load().mnf_pack_gather(
                    self._flat.data_ptr(), self._index.data_ptr(), image.data_ptr(),
                    self._index.numel(), _stream()))
                self._image = image
            else:
                self._image = None
            self._device_split_index(device)
            if self._split_index:
                sidx, n_split, n_plain = self._split_index
                split = torch.empty(n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32, device=device)
                _lib.check("mnf_pack_gather_split", _lib.load().mnf_pack_gather_split(
                    self._flat.data_ptr(), sidx.data_ptr(), split.data_ptr(), n_split, n_plain, _stream()))
                self._split = split
            else:
                self._split = None
            self.__dict__["_images_key"] = key
        return self._flat, self._image

    # out-of-place layer on raw buffers; accum: (rows,) tensor receiving ``+= log_det`` or None
    def _run(self, x: Tensor, inverse: bool, accum: Tensor | None) -> tuple[Tensor, Tensor | None]:
        raise NotImplementedError



class _TwoWayFlow(_HipFlow):
    """Flows with both directions (everything except the forward-only RNVP)."""

    def forward(self, z: Tensor) -> tuple[Tensor, Tensor]:
        return self._run(z, False, None)

    def inverse(self, x: Tensor) -> tuple[Tensor, Tensor]:
        return self._run(x, True, None)


class AffineHalfFlow(_TwoWayFlow):
    """RealNVP / NICE half coupling (flows/affine_half_flow.py:20-66)."""

    def __init__(self, dim: int, parity: bool, h_sizes: Sequence[int] = (24, 24, 24),
                 scale: bool = True, shift: bool = True) -> None:
        super().__init__()
        if dim % 2:
            raise ValueError("AffineHalfFlow needs an even dim")
        self.dim = int(dim)
        self.parity = parity
        self.h_sizes = tuple(int(h) for h in h_sizes)
        self.scale, self.shift = bool(scale), bool(shift)
        # absent nets return zeros in the reference (:38); here the kernels take flags
        if scale:
            self.s_net = MLP(dim // 2, *self.h_sizes, dim // 2)
        if shift:
            self.t_net = MLP(dim // 2, *self.h_sizes, dim // 2)
        self._hid = _lib.int_array(self.h_sizes)
        self._split_ok = not _narrow_hidden(self.h_sizes)

    def _packed_params(self) -> list[Tensor]:
        return self._net_params(([self.s_net] if self.scale else []) + ([self.t_net] if self.shift else []))

    def _image_index_host(self):
        lib = _lib.load()
        n = lib.mnf_affine_half_image_floats(self.dim, len(self.h_sizes), self._hid, self.scale, self.shift)
        if n <= 0:
            return None
        idx = (ctypes.c_int32 * n)()
        _lib.check("mnf_affine_half_image_index", lib.mnf_affine_half_image_index(
            self.dim, len(self.h_sizes), self._hid, self.scale, self.shift, idx))
        return idx

    def _bwd_split_index(self, device):
        """(device index table, n_split_words, n_plain_words) of the split gradient kernel's operand image (forward
        and transposed weights), built once per module; False: no such kernel for this shape."""
        cached = self.__dict__.get("_bwd_split_index_cache")
        if cached is None or (cached and cached[0].device != device):
            lib = _lib.load()
            n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
            rc = lib.mnf_affine_half_bwd_split_layout(self.dim, len(self.h_sizes), self._hid, self.scale, self.shift,
                                                      ctypes.byref(n_split), ctypes.byref(n_plain))
            cached = False
            if rc != _lib.MNF_ERR_UNSUPPORTED:
                _lib.check("mnf_affine_half_bwd_split_layout", rc)
                idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
                _lib.check("mnf_affine_half_bwd_split_index", lib.mnf_affine_half_bwd_split_index(
                    self.dim, len(self.h_sizes), self._hid, self.scale, self.shift, idx))
                cached = (torch.frombuffer(idx, dtype=torch.int32).clone().to(device), n_split.value, n_plain.value)
            self.__dict__["_bwd_split_index_cache"] = cached
        return cached

    def _bwd_split_ok(self) -> bool:
        return not (self.force_fp32_mfma or _FP32_MFMA_ENV or _dispatch.BWD_FP32 or self.force_generic or not self._split_ok)

    def _bwd_split_image(self, device, flat: Tensor) -> Tensor | None:
        """The split gradient kernel's operand image for the parameters in ``flat`` (packed per call: the weights
        change between training steps), or None when this layer's gradients run on the fp32 kernels."""
        if not self._bwd_split_ok():
            return None
        table = self._bwd_split_index(device)
        if not table:
            return None
        idx, n_split, n_plain = table
        image = torch.empty(n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32, device=device)
        _lib.check("mnf_pack_gather_split", _lib.load().mnf_pack_gather_split(
            flat.data_ptr(), idx.data_ptr(), image.data_ptr(), n_split, n_plain, _stream()))
        return image

    def _bwd_index(self, device) -> Tensor | None:
        """Device index table of the MFMA gradient kernel (built once per module), or None."""
        cached = self.__dict__.get("_bwd_index_cache")
        if cached is None or (cached[0] is not None and cached[0].device != device):
            lib = _lib.load()
            n = lib.mnf_affine_half_bwd_index_ints(self.dim, len(self.h_sizes), self._hid, self.scale, self.shift)
            table = None
            if n > 0:
                idx = (ctypes.c_int32 * n)()
                _lib.check("mnf_affine_half_bwd_index", lib.mnf_affine_half_bwd_index(
                    self.dim, len(self.h_sizes), self._hid, self.scale, self.shift, idx))
                table = torch.frombuffer(idx, dtype=torch.int32).clone().to(device)
            cached = (table,)
            self.__dict__["_bwd_index_cache"] = cached
        return cached[0]

    def _split_index_host(self):
        lib = _lib.load()
        n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = lib.mnf_affine_half_split_layout(self.dim, len(self.h_sizes), self._hid, self.scale, self.shift,
                                              ctypes.byref(n_split), ctypes.byref(n_plain))
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            return None
        _lib.check("mnf_affine_half_split_layout", rc)
        idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
        _lib.check("mnf_affine_half_split_index", lib.mnf_affine_half_split_index(
            self.dim, len(self.h_sizes), self._hid, self.scale, self.shift, idx))
        return idx, n_split.value, n_plain.value

    def _run(self, x, inverse, accum, sqnorm: Tensor | None = None, overwrite: bool = False):
        # overwrite: accum receives ld instead of += ld (first layer of a pass: saves zero-filling it)
        if accum is None and sqnorm is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 \
                and _wants_grad(self, x):
            xg = _grad_input(x)
            if xg.shape[1] != self.dim:
                raise ValueError(f"expected dim {self.dim}, got {xg.shape[1]}")
            params = self._packed_params()
            flat = torch.cat([p.reshape(-1) for p in params]) if params else xg.new_empty(0)
            return _AffineHalfFn.apply(xg, flat, self, bool(inverse))
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        if x.shape[0] == 0:
            return _empty_result(x, accum)
        flat, image, split = self._packed3(x.device)
        y = torch.empty_like(x)
        ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_affine_half_sq", _lib.load().mnf_affine_half_sq(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), _ptr(sqnorm), int(accum is not None and not overwrite),
            _ptr(flat), _ptr(image), _ptr(split), x.shape[0], self.dim, int(bool(self.parity)),
            int(inverse), len(self.h_sizes),
            self._hid, int(self.scale), int(self.shift), self._force_code(image), _stream()))
        if not self.force_generic:
            _lib.note_generic("AffineHalfFlow", x.shape[0], f"dim={self.dim}, hidden={self.h_sizes}")
        return y, (None if accum is not None else ld)

    def emits_sqnorm(self, device) -> bool:
        """True when this layer's kernel can also write |y_row|^2 (the specialised kernels only; a half narrower than
        its MFMA tile only has the split stack kernel, so not when that one is switched off)."""
        if self.force_generic:
            return False
        _, image, split = self._packed3(device)
        return image is not None and (split is not None or self.dim // 2 in (16, 32, 64, 128))

    def forward(self, z: Tensor, inverse: bool = False) -> tuple[Tensor, Tensor]:
        return self._run(z, inverse, None)


class NSF_CL(_TwoWayFlow):
    """Neural-spline coupling layer (flows/spline_flow.py:238-285)."""

    def __init__(self, dim: int, K: int = 5, B: float = 3, n_h: int = 8, net_class=MLP) -> None:
        super().__init__()
        if dim % 2:
            raise ValueError("NSF_CL needs an even dim")
        self.dim, self.K, self.B = int(dim), int(K), B
        self.f1 = net_class(dim // 2, n_h, n_h, n_h, (3 * K - 1) * dim // 2)
        self.f2 = net_class(dim // 2, n_h, n_h, n_h, (3 * K - 1) * dim // 2)
        _require_mlp(self.f1, self.f2)
        sizes = self.f1.layer_sizes
        self.h_sizes = tuple(int(s) for s in sizes[1:-1])
        self._hid = _lib.int_array(self.h_sizes)
        self._split_ok = not _narrow_hidden(self.h_sizes)

    def _packed_params(self) -> list[Tensor]:
        return self._net_params([self.f1, self.f2])

    def _image_index_host(self):
        lib = _lib.load()
        n = lib.mnf_nsf_cl_image_floats(self.dim, self.K, len(self.h_sizes), self._hid)
        if n <= 0:
            return None
        idx = (ctypes.c_int32 * n)()
        _lib.check("mnf_nsf_cl_image_index", lib.mnf_nsf_cl_image_index(
            self.dim, self.K, len(self.h_sizes), self._hid, idx))
        return idx

    def _split_index_host(self):
        lib = _lib.load()
        n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = lib.mnf_nsf_cl_split_layout(self.dim, self.K, len(self.h_sizes), self._hid, ctypes.byref(n_split),
                                         ctypes.byref(n_plain))
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            return None
        _lib.check("mnf_nsf_cl_split_layout", rc)
        idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
        _lib.check("mnf_nsf_cl_split_index", lib.mnf_nsf_cl_split_index(
            self.dim, self.K, len(self.h_sizes), self._hid, idx))
        return idx, n_split.value, n_plain.value

    def _bwd_tile_tables(self, device):
        """(operand index table, flush table, n_split_words, n_plain_words) of the tile gradient kernel on the device,
        built once per module; False: no such kernel for this shape."""
        cached = self.__dict__.get("_bwd_tile_cache")
        if cached is None or (cached and cached[0].device != device):
            lib = _lib.load()
            n_split, n_plain, n_params = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
            rc = lib.mnf_nsf_cl_bwd_tile_layout(self.dim, self.K, len(self.h_sizes), self._hid, ctypes.byref(n_split),
                                                ctypes.byref(n_plain), ctypes.byref(n_params))
            cached = False
            if rc != _lib.MNF_ERR_UNSUPPORTED:
                _lib.check("mnf_nsf_cl_bwd_tile_layout", rc)
                idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
                flush = (ctypes.c_int32 * n_params.value)()
                _lib.check("mnf_nsf_cl_bwd_tile_index", lib.mnf_nsf_cl_bwd_tile_index(
                    self.dim, self.K, len(self.h_sizes), self._hid, idx, flush))
                cached = (torch.frombuffer(idx, dtype=torch.int32).clone().to(device),
                          torch.frombuffer(flush, dtype=torch.int32).clone().to(device), n_split.value, n_plain.value)
            self.__dict__["_bwd_tile_cache"] = cached
        return cached

    # ---- halves that are not whole float4 groups (dim = 2, 4, 6, 10, ...; the reference's own test shape is dim = 2,
    # tests/test_flows.py:89-99).  The matrix-core kernels take halves in whole groups of four columns, so such a layer
    # runs them on a padded TWIN: each half widened to the next multiple of four, the conditioners' first / last layers
    # zero-padded to match (a padded column meets zero weights; a padded element has no parameters), and the padded
    # columns of x set beyond the tail bound, where the spline is the identity with log-derivative 0
    # (spline_flow.py:72-85) -- so y, log_det and every gradient of the real columns and parameters are what the layer
    # itself would give, and autograd's own pad / slice nodes carry the gradients back.  From _dispatch.NSF_PAD_MIN_ROWS rows on
    # (below that the extra pad / slice launches cost more than the any-shape kernels do).
    def _pad_half(self) -> int:
        """Padded half width of the twin (0: none -- the halves are whole groups already, or no kernel takes the twin)."""
        cached = self.__dict__.get("_pad_half_cache")
        if cached is None:
            hr = self.dim // 2
            hp = (hr + 3) // 4 * 4
            cached = 0
            if hp != hr and 2 * hp <= 64 and not _narrow_hidden(self.h_sizes) and len(set(self.h_sizes)) == 1:
                if _lib.load().mnf_nsf_cl_bwd_tile_supported(2 * hp, self.K, len(self.h_sizes), self._hid):
                    cached = hp
            self.__dict__["_pad_half_cache"] = cached
        return cached

    def _padded_params(self, hp: int) -> list[Tensor]:
        """This layer's parameters in the twin's shapes (differentiable: F.pad nodes)."""
        hr, P = self.dim // 2, 3 * self.K - 1
        out = []
        for net in (self.f1, self.f2):
            ps = self._net_params([net])
            for i, prm in enumerate(ps):
                if i == 0:  # first layer's weight (n_h, hr): zero columns for the padded inputs
                    prm = torch.nn.functional.pad(prm, (0, hp - hr))
                elif i == len(ps) - 2:  # last layer's weight ((3K-1) hr, n_h), element-major rows: none for padded elements
                    prm = torch.nn.functional.pad(prm, (0, 0, 0, P * (hp - hr)))
                elif i == len(ps) - 1:  # last layer's bias
                    prm = torch.nn.functional.pad(prm, (0, P * (hp - hr)))
                out.append(prm)
        return out

    def _twin(self, device, hp: int) -> "NSF_CL":
        """The padded layer, its parameters refreshed from this layer's whenever they changed."""
        twin = self.__dict__.get("_twin_module")
        if twin is None or twin.f1[0].weight.device != device:
            twin = NSF_CL(2 * hp, K=self.K, B=self.B, n_h=self.h_sizes[0]).to(device).requires_grad_(False)
            self.__dict__["_twin_module"] = twin
            self.__dict__["_twin_key"] = None
        params = self._packed_params()
        key = (_flat_gen(self), tuple((q.data_ptr(), q._version) for q in params))
        if key != self.__dict__.get("_twin_key"):
            with torch.no_grad():
                for dst, src in zip(twin._packed_params(), self._padded_params(hp)):
                    dst.copy_(src)
            self.__dict__["_twin_key"] = key
        twin.force_fp32_mfma = self.force_fp32_mfma
        return twin

    def _run_padded(self, x, inverse, accum, hp: int):
        hr, rows = self.dim // 2, x.shape[0]
        twin = self._twin(x.device, hp)
        fill = x.new_full((rows, hp - hr), 2.0 * float(self.B) + 1.0)  # outside [-B, B]: identity, log-derivative 0
        wants = accum is None and _wants_grad(self, x)
        xs = _grad_input(x) if wants else _device_input(x, "input")
        x_pad = torch.cat([xs[:, :hr], fill, xs[:, hr:], fill], dim=1)
        if wants:
            flat_pad = torch.cat([q.reshape(-1) for q in self._padded_params(hp)])
            y_pad, ld = _NsfFn.apply(x_pad, flat_pad, twin, bool(inverse))
        else:
            with torch.no_grad():
                y_pad, ld = twin._run(x_pad, inverse, accum)
        return torch.cat([y_pad[:, :hr], y_pad[:, hp:hp + hr]], dim=1), ld

    def _run(self, x, inverse, accum):
        if (isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[1] == self.dim and x.dtype == torch.float32
                and x.shape[0] >= _dispatch.NSF_PAD_MIN_ROWS and not self.force_generic):
            hp = self._pad_half()
            if hp:
                return self._run_padded(x, inverse, accum, hp)
        if accum is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 and _wants_grad(self, x):
            xg = _grad_input(x)
            if xg.shape[1] != self.dim:
                raise ValueError(f"expected dim {self.dim}, got {xg.shape[1]}")
            flat = torch.cat([p.reshape(-1) for p in self._packed_params()])
            return _NsfFn.apply(xg, flat, self, bool(inverse))
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        if x.shape[0] == 0:
            return _empty_result(x, accum)
        flat, image, split = self._packed3(x.device)
        y = torch.empty_like(x)
        ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_nsf_cl", _lib.load().mnf_nsf_cl(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), int(accum is not None), _ptr(flat), _ptr(image),
            _ptr(split),
            x.shape[0], self.dim, self.K, float(self.B), int(inverse), len(self.h_sizes), self._hid,
            self._force_code(image), _stream()))
        if not self.force_generic and not self._pad_half():  # (a padded-twin shape is here by choice: few rows)
            _lib.note_generic("NSF_CL", x.shape[0], f"dim={self.dim}, K={self.K}, hidden={self.h_sizes}")
        return y, (None if accum is not None else ld)


class _NsfArFn(torch.autograd.Function):
    """NSF_AR with gradients (mnf_nsf_ar_bwd: recompute, then reverse mode through splines and conditioners)."""

    @staticmethod
    def forward(ctx, x, flat_with_grad, module, inverse):
        flat = flat_with_grad.detach().contiguous()
        y = torch.empty_like(x)
        ld = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_nsf_ar", _lib.load().mnf_nsf_ar(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), 0, flat.data_ptr(), x.shape[0], module.dim, module.K,
            float(module.B), int(inverse), len(module.h_sizes), module._hid, _stream()))
        ctx.module, ctx.inverse = module, inverse
        ctx.save_for_backward(x, flat)
        return y, ld

    @staticmethod
    def backward(ctx, grad_y, grad_ld):
        x, flat = ctx.saved_tensors
        m = ctx.module
        gy = None if grad_y is None else grad_y.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        grad_x = torch.empty_like(x)
        grad_flat = torch.zeros_like(flat)
        _lib.check("mnf_nsf_ar_bwd", _lib.load().mnf_nsf_ar_bwd(
            x.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(), x.shape[0],
            m.dim, m.K, float(m.B), int(ctx.inverse), len(m.h_sizes), m._hid, _stream()))
        return grad_x, grad_flat, None, None


class NSF_AR(_TwoWayFlow):
    """Neural-spline autoregressive layer (flows/spline_flow.py:182-235): element i is moved by a spline
    parametrised by ``layers[i-1](first i elements)`` -- of the output in ``forward`` (sequential), of the input in
    ``inverse`` --, element 0 by ``init_param``.  Same constructor, attribute names and state_dict keys
    (``init_param``, ``layers.{i}.{0,2,4,6}.{weight,bias}``) as the reference; the arithmetic is one
    ``mnf_nsf_ar`` launch per direction on the spline device function NSF_CL uses."""

    def __init__(self, dim: int, K: int = 5, B: float = 3, n_h: int = 8, net_class=MLP) -> None:
        super().__init__()
        self.dim, self.K, self.B = int(dim), int(K), B
        self.layers = nn.ModuleList()
        self.init_param = nn.Parameter(torch.empty(3 * K - 1))
        for i in range(1, dim):
            self.layers.append(net_class(i, n_h, n_h, n_h, 3 * K - 1))
        _require_mlp(*self.layers)
        self.h_sizes = tuple(int(s) for s in self.layers[0].layer_sizes[1:-1]) if dim > 1 else (n_h, n_h, n_h)
        self._hid = _lib.int_array(self.h_sizes)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.uniform_(self.init_param, -1 / 2, 1 / 2)  # spline_flow.py:196-197

    def _packed_params(self) -> list[Tensor]:
        return [self.init_param] + self._net_params(list(self.layers))

    def _run(self, x, inverse, accum):
        if accum is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 and _wants_grad(self, x):
            xg = _grad_input(x)
            if xg.shape[1] != self.dim:
                raise ValueError(f"expected dim {self.dim}, got {xg.shape[1]}")
            flat = torch.cat([p.reshape(-1) for p in self._packed_params()])
            return _NsfArFn.apply(xg, flat, self, bool(inverse))
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        if x.shape[0] == 0:
            return _empty_result(x, accum)
        flat, _ = self._packed(x.device)
        y = torch.empty_like(x)
        ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_nsf_ar", _lib.load().mnf_nsf_ar(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), int(accum is not None), flat.data_ptr(), x.shape[0], self.dim,
            self.K, float(self.B), int(inverse), len(self.h_sizes), self._hid, _stream()))
        return y, (None if accum is not None else ld)


def rqs(inputs: Tensor, W: Tensor, H: Tensor, D: Tensor, inverse: bool = False,
        tail_bound: float = 1.0) -> tuple[Tensor, Tensor]:
    """``unconstrained_RQS`` (flows/spline_flow.py:29-68) elementwise on the GPU.

    inputs (...,), W/H (..., K), D (..., K-1).  An all-outside batch returns the identity
    (the reference raises there)."""
    shape = inputs.shape
    K = W.shape[-1]
    v = _device_input(inputs.reshape(-1, 1), "inputs").reshape(-1)
    w = _device_input(W.reshape(-1, K), "W")
    h = _device_input(H.reshape(-1, K), "H")
    d = _device_input(D.reshape(-1, max(K - 1, 1)) if K > 1 else D.reshape(-1, 1), "D")
    out, lad = torch.empty_like(v), torch.empty_like(v)
    _lib.check("mnf_rqs", _lib.load().mnf_rqs(
        v.data_ptr(), w.data_ptr(), h.data_ptr(), d.data_ptr(), out.data_ptr(), lad.data_ptr(),
        v.numel(), K, float(tail_bound), int(inverse), _stream()))
    return out.reshape(shape), lad.reshape(shape)


_RNVP_BWD_WORK: dict = {}  # (device, stream) -> scratch of mnf_rnvp_bwd_mfma (two streams never share hand-over tiles)


class RNVP(_HipFlow):
    """Forward-only masked/gated coupling used by the MNF layers (flows/rnvp.py:7-39).

    ``forward(z)`` uses a fresh Bernoulli(0.5) mask per element per call like the reference (:28);
    the bits come from the library's counter-based generator, keyed by a seed drawn from torch's
    global RNG (so ``torch.manual_seed`` makes runs reproducible) and never touch memory.
    ``forward(z, mask=m)`` takes an explicit float mask; ``forward(z, seed=k)`` a fixed seed, and
    ``mask_for(seed, rows)`` returns the mask such a call used."""

    def __init__(self, dim: int, h_sizes: Sequence[int] = (30,)) -> None:
        super().__init__()
        self.dim = int(dim)
        self.h_sizes = tuple(int(h) for h in h_sizes)
        self.net = MLP(dim, *self.h_sizes)
        self.t = nn.Linear(self.h_sizes[-1], dim)
        self.s = nn.Linear(self.h_sizes[-1], dim)
        self._hid = _lib.int_array(self.h_sizes)

    def _packed_params(self) -> list[Tensor]:
        return self._net_params([self.net, self.t, self.s])

    def _image_index_host(self):
        lib = _lib.load()
        n = lib.mnf_rnvp_image_floats(self.dim, len(self.h_sizes), self._hid)
        if n <= 0:
            return None
        idx = (ctypes.c_int32 * n)()
        _lib.check("mnf_rnvp_image_index", lib.mnf_rnvp_image_index(self.dim, len(self.h_sizes), self._hid, idx))
        return idx

    def _split_index_host(self):
        lib = _lib.load()
        n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = lib.mnf_rnvp_split_layout(self.dim, len(self.h_sizes), self._hid, ctypes.byref(n_split),
                                       ctypes.byref(n_plain))
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            return None
        _lib.check("mnf_rnvp_split_layout", rc)
        idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
        _lib.check("mnf_rnvp_split_index", lib.mnf_rnvp_split_index(self.dim, len(self.h_sizes), self._hid, idx))
        return idx, n_split.value, n_plain.value

    def _bwd_index(self, device):
        """(device index table, n_split_words, n_plain_words) of the gradient kernels' operand image, or False."""
        cached = self.__dict__.get("_bwd_idx")
        if cached is None or (cached and cached[0].device != device):
            lib = _lib.load()
            n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
            rc = lib.mnf_rnvp_bwd_mfma_layout(self.dim, len(self.h_sizes), self._hid, ctypes.byref(n_split),
                                              ctypes.byref(n_plain))
            if rc == _lib.MNF_ERR_UNSUPPORTED:
                cached = False
            else:
                _lib.check("mnf_rnvp_bwd_mfma_layout", rc)
                idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
                _lib.check("mnf_rnvp_bwd_mfma_index", lib.mnf_rnvp_bwd_mfma_index(self.dim, len(self.h_sizes), self._hid, idx))
                cached = (torch.frombuffer(idx, dtype=torch.int32).clone().to(device), n_split.value, n_plain.value)
            self.__dict__["_bwd_idx"] = cached
        return cached

    def _bwd_image(self, device, flat: Tensor) -> Tensor | None:
        """The gradient kernels' operand image for the parameters in ``flat`` (repacked per backward pass: the weights
        change between steps, and the pack is one small launch), or None when the shape has no such kernels."""
        if self.force_fp32_mfma or _FP32_MFMA_ENV or not self._split_ok:
            return None
        table = self._bwd_index(device)
        if not table:
            return None
        idx, n_split, n_plain = table
        # one pack per parameter state: `flat` is the forward pass's packed-parameter tensor (replaced, never rewritten,
        # when a parameter changes), and a layer is differentiated several times per step (MNF: sample_z in forward
        # and again in kl_div)
        cached = self.__dict__.get("_bwd_img")
        capturing = torch.cuda.is_current_stream_capturing()
        if cached is not None and cached[0] is flat and cached[1] == (flat._version, capturing):
            return cached[2]
        image = torch.empty(n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32, device=device)
        _lib.check("mnf_pack_gather_split", _lib.load().mnf_pack_gather_split(
            flat.data_ptr(), idx.data_ptr(), image.data_ptr(), n_split, n_plain, _stream()))
        self.__dict__["_bwd_img"] = (flat, (flat._version, capturing), image)
        return image

    def _bwd_workspace(self, lib, rows: int, device) -> Tensor:
        """Scratch of the gradient kernels (1 KB per row + flags).  One buffer per (device, stream), shared by every RNVP
        layer and grown on demand: the layers' backward passes follow one another on a stream; two streams (a side-stream
        warm-up next to eager work, a captured graph's private pool) never share hand-over tiles."""
        need = int(lib.mnf_rnvp_bwd_mfma_workspace_bytes(rows, self.dim, len(self.h_sizes), self._hid))
        key = (device, _stream())
        work = _RNVP_BWD_WORK.get(key)
        if work is None or work.numel() < need:
            work = torch.empty(need, dtype=torch.uint8, device=device)
            _RNVP_BWD_WORK[key] = work
        return work

    def _few(self, rows: int, explicit_mask: bool) -> bool:
        """Does the library run a forward call of this many rows on the few-rows kernel (mnf_rnvp_few.hip)?  It reads
        the plain parameter buffer: no operand image is needed then."""
        if self.force_generic:
            return False
        cache = self.__dict__.setdefault("_few_cache", {})
        ok = cache.get((rows, explicit_mask))
        if ok is None:
            ok = cache[(rows, explicit_mask)] = bool(_lib.load().mnf_rnvp_few_rows_ok(
                rows, self.dim, len(self.h_sizes), self._hid, int(explicit_mask)))
        return ok

    def mask_for(self, seed: int, rows: int, device="cuda") -> Tensor:
        m = torch.empty(rows, self.dim, dtype=torch.float32, device=device)
        if rows:
            _lib.check("mnf_rnvp_mask", _lib.load().mnf_rnvp_mask(int(seed), m.data_ptr(), rows, self.dim, _stream()))
        return m

    def _run(self, z, inverse, accum, mask: Tensor | None = None, seed: int | None = None, prologue=None):
        """prologue = (q0_mean, q0_log_var): ``z`` holds eps and the layer runs on q0_mean + q0_std * eps formed
        inside the kernel (MNFLinear.sample_z); returns None when that fused form is not available."""
        if inverse:
            raise AttributeError("RNVP has no inverse (flows/rnvp.py defines forward only)")
        want_grad = accum is None and isinstance(z, Tensor) and z.is_cuda and z.shape[0] > 0 and _wants_grad(self, z)
        zin = z
        z = _device_input(z, "input")
        if z.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {z.shape[1]}")
        if z.shape[0] == 0:
            return _empty_result(z, accum)
        if mask is not None:
            mask = _device_input(mask, "mask")
            if mask.shape != z.shape:
                raise ValueError("mask must have the shape of z")
        elif seed is None and _DEVICE_MASKS:
            # the reference's own draw (rnvp.py:28), on the device generator: unlike a host-drawn seed (a kernel
            # argument) it is redrawn by every replay of a captured hipGraph
            mask = torch.empty_like(z).bernoulli_(0.5)  # (one kernel; full_like + bernoulli would be two)
        elif seed is None:  # one draw from torch's global generator per call
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        if want_grad and prologue is not None:
            return None
        if want_grad:
            params = self._packed_params()
            home = _flat_home_of(self, params)
            flat_g = _home_stand_in(self, z.device) if home is not None else torch.cat([p.reshape(-1) for p in params])
            return _RnvpFn.apply(_grad_input(zin), flat_g, self, mask, int(seed or 0) & 0xFFFFFFFFFFFFFFFF, home)
        few = prologue is None and self._few(z.shape[0], mask is not None)
        if few:
            flat, image, split = self._packed(z.device, images=False)[0], None, None
        else:
            flat, image, split = self._packed3(z.device)
        x = torch.empty_like(z)
        ld = accum if accum is not None else torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        if prologue is not None:
            if image is None or split is None or self.force_generic:
                return None
            mean, log_var = prologue
            rc = _lib.load().mnf_rnvp_sample(
                z.data_ptr(), mean.data_ptr(), log_var.data_ptr(), _ptr(mask), int(seed or 0) & 0xFFFFFFFFFFFFFFFF,
                x.data_ptr(), ld.data_ptr(), int(accum is not None), image.data_ptr(), split.data_ptr(), z.shape[0],
                self.dim, len(self.h_sizes), self._hid, _stream())
            if rc == _lib.MNF_ERR_UNSUPPORTED:
                return None
            _lib.check("mnf_rnvp_sample", rc)
            return x, (None if accum is not None else ld)
        _lib.check("mnf_rnvp_seeded", _lib.load().mnf_rnvp_seeded(
            z.data_ptr(), _ptr(mask), int(seed or 0) & 0xFFFFFFFFFFFFFFFF, x.data_ptr(), ld.data_ptr(),
            int(accum is not None), _ptr(flat), _ptr(image), _ptr(split), z.shape[0], self.dim,
            len(self.h_sizes), self._hid, int(self.force_generic) if few else self._force_code(image), _stream()))
        if not self.force_generic:
            _lib.note_generic("RNVP", z.shape[0], f"dim={self.dim}, hidden={self.h_sizes}")
        return x, (None if accum is not None else ld)

    def forward(self, z: Tensor, mask: Tensor | None = None, seed: int | None = None) -> tuple[Tensor, Tensor]:
        return self._run(z, False, None, mask, seed)


class _MafFn(torch.autograd.Function):
    """MAF / IAF with gradients (mnf_maf / mnf_maf_bwd).  ``home``: see _RnvpFn."""

    @staticmethod
    def forward(ctx, x, flat_with_grad, module, sequential, home=None):
        flat, _ = module._packed(x.device)
        masks = module._mask_bytes(x.device)
        y = torch.empty_like(x)
        ld = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_maf", _lib.load().mnf_maf(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), 0, flat.data_ptr(), masks.data_ptr(), x.shape[0], module.dim,
            int(bool(module.parity)), int(sequential), len(module.h_sizes), module._hid, _stream()))
        ctx.module, ctx.sequential, ctx.home = module, sequential, home
        ctx.save_for_backward(x, y, flat, masks)
        return y, ld

    @staticmethod
    def backward(ctx, grad_y, grad_ld):
        x, y, flat, masks = ctx.saved_tensors
        m, home = ctx.module, ctx.home
        gy = None if grad_y is None else grad_y.contiguous()
        gl = None if grad_ld is None else grad_ld.contiguous()
        grad_x = torch.empty_like(x)
        if home is not None:
            grad_flat, ret = home[0].grad[home[1]:home[1] + home[2]], None
        else:
            grad_flat = ret = torch.zeros_like(flat)
        _lib.check("mnf_maf_bwd", _lib.load().mnf_maf_bwd(
            x.data_ptr(), y.data_ptr(), _ptr(gy), _ptr(gl), grad_x.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(),
            masks.data_ptr(), x.shape[0], m.dim, int(bool(m.parity)), int(ctx.sequential), len(m.h_sizes), m._hid,
            _stream()))
        return grad_x, ret, None, None, None


class MAF(_TwoWayFlow):
    """Masked autoregressive flow (flows/maf.py:21-62): ``inverse`` is one pass of the MADE network (density
    estimation), ``forward`` decodes the elements one at a time (dim passes).  Same constructor, attribute names and
    state_dict keys (``net.{2l}.weight / .bias / .mask``) as the reference; each direction is one ``mnf_maf`` launch
    (generic path: a thread per row, masked weights in LDS), gradients from ``mnf_maf_bwd``.  ``net`` must be a
    ``MADE(dim, hidden, 2 * dim)``: the kernels evaluate the masked network themselves."""

    _sequential_forward = True  # IAF: the two directions swapped

    def __init__(self, dim: int, parity: bool, net: nn.Module | None = None, h_sizes: Sequence[int] = (24, 24, 24)) -> None:
        super().__init__()
        self.dim, self.parity = int(dim), parity
        self.net = net or MADE(dim, h_sizes, 2 * dim, natural_ordering=True)
        if not isinstance(self.net, MADE) or self.net.n_in != self.dim or self.net.n_out != 2 * self.dim:
            raise NotImplementedError(
                f"torch_mnf_amd's MAF / IAF kernels evaluate a MADE({self.dim}, hidden, {2 * self.dim}) network themselves; "
                f"got {type(self.net).__name__}")
        self.h_sizes = tuple(int(h) for h in self.net.hidden_sizes)
        self._hid = _lib.int_array(self.h_sizes)

    def _masked(self) -> list:
        return [m for m in self.net if isinstance(m, MaskedLinear)]

    def _packed_params(self) -> list[Tensor]:
        return [p for m in self._masked() for p in (m.weight, m.bias)]

    def _mask_bytes(self, device) -> Tensor:
        """The MaskedLinear mask buffers as bytes, back to back; rebuilt when a mask tensor is replaced
        (``MADE.update_masks`` assigns new ones) or moved."""
        masks = [m.mask for m in self._masked()]
        key = (device, tuple((id(t), t._version) for t in masks))
        cached = self.__dict__.get("_mask_cache")
        if cached is None or cached[0] != key:
            packed = torch.cat([(t != 0).reshape(-1).to(device=device, dtype=torch.uint8) for t in masks]).contiguous()
            cached = self.__dict__["_mask_cache"] = (key, packed)
        return cached[1]

    def _autoregressive_in_index_order(self) -> bool:
        """Does output i (and dim + i) of the masked network depend on inputs j < i only?  (The composed masks are
        strictly lower triangular: MADE's natural ordering.)  Cached with the mask tensors' identities and versions."""
        masks = [m.mask for m in self._masked()]
        key = tuple((id(t), t._version) for t in masks)
        cached = self.__dict__.get("_order_cache")
        if cached is None or cached[0] != key:
            conn = None  # (outputs of the layers so far, inputs) connectivity
            for t in masks:
                layer = (t.detach().to("cpu", torch.float64) != 0).to(torch.float64).T  # MaskedLinear.mask is (in, out)
                conn = layer if conn is None else ((layer @ conn) != 0).to(torch.float64)
            out_elem = torch.arange(conn.shape[0]) % self.dim
            allowed = torch.arange(self.dim)[None, :] < out_elem[:, None]
            cached = self.__dict__["_order_cache"] = (key, bool(((conn != 0) & ~allowed).sum() == 0))
        return cached[1]

    def _run(self, x, inverse, accum):
        sequential = bool(inverse) != self._sequential_forward
        if accum is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 and _wants_grad(self, x):
            if sequential and not self._autoregressive_in_index_order():
                # mnf_maf_bwd evaluates the network ONCE on the decoded output and reuses those activations for every
                # step of the element-by-element pass: right only when element i never sees elements >= i, i.e. for
                # masks that are autoregressive in index order.  With any other MADE (natural_ordering=False, several
                # masks, masks from a state_dict) the VALUES of this direction still match the reference (flows/maf.py:
                # 39-50: elements not yet decoded are zero at step i), its gradients would not.
                raise NotImplementedError(
                    f"{type(self).__name__}: gradients of the element-by-element direction need a MADE whose masks are "
                    f"autoregressive in index order (MADE(..., natural_ordering=True)); the one-pass direction and both "
                    f"directions' values work with any MADE")
            xg = _grad_input(x)
            if xg.shape[1] != self.dim:
                raise ValueError(f"expected dim {self.dim}, got {xg.shape[1]}")
            params = self._packed_params()
            home = _flat_home_of(self, params)
            flat = _home_stand_in(self, xg.device) if home is not None else torch.cat([p.reshape(-1) for p in params])
            return _MafFn.apply(xg, flat, self, sequential, home)
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        if x.shape[0] == 0:
            return _empty_result(x, accum)
        flat, _ = self._packed(x.device)
        masks = self._mask_bytes(x.device)
        y = torch.empty_like(x)
        ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check("mnf_maf", _lib.load().mnf_maf(
            x.data_ptr(), y.data_ptr(), ld.data_ptr(), int(accum is not None), flat.data_ptr(), masks.data_ptr(),
            x.shape[0], self.dim, int(bool(self.parity)), int(sequential), len(self.h_sizes), self._hid, _stream()))
        return y, (None if accum is not None else ld)


class IAF(MAF):
    """Inverse autoregressive flow (flows/maf.py:65-72): MAF with ``forward`` and ``inverse`` swapped -- one pass to
    sample, dim passes to evaluate a density."""

    _sequential_forward = False


class AffineConstantFlow(_TwoWayFlow):
    """Per-dimension learned affine; log_det has shape (1,) (flows/affine_constant_flow.py:7-26)."""

    def __init__(self, dim: int, scale: bool = True, shift: bool = True) -> None:
        super().__init__()
        self.dim = int(dim)
        if scale:
            self.s = nn.Parameter(torch.randn(1, dim))
        else:
            self.register_buffer("s", torch.zeros(1, dim), persistent=False)
        if shift:
            self.t = nn.Parameter(torch.randn(1, dim))
        else:
            self.register_buffer("t", torch.zeros(1, dim), persistent=False)

    def _run(self, x, inverse, accum):
        if accum is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 and _wants_grad(self, x):
            xg = _grad_input(x)
            y = _AffineConstFn.apply(xg, self.s.to(xg.device), self.t.to(xg.device), bool(inverse))
            return y, torch.sum(-self.s if inverse else self.s, dim=1).to(xg.device)
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        s = self.s.detach().to(x.device, torch.float32).contiguous()
        t = self.t.detach().to(x.device, torch.float32).contiguous()
        if x.shape[0] == 0:
            ld1 = (-s if inverse else s).sum(dim=1)
            return torch.empty_like(x), (None if accum is not None else ld1)
        y = torch.empty_like(x)
        ld1 = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check("mnf_affine_const", _lib.load().mnf_affine_const(
            x.data_ptr(), y.data_ptr(), s.data_ptr(), t.data_ptr(), _ptr(accum), int(accum is not None),
            ld1.data_ptr(), x.shape[0], self.dim, int(inverse), _stream()))
        return y, (None if accum is not None else ld1)


class ActNormFlow(AffineConstantFlow):
    """AffineConstantFlow with a data-dependent initialisation on the first ``inverse`` call
    (flows/affine_constant_flow.py:29-50): s <- log std(x, 0), t <- mean(x * exp(s), 0)."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.data_dep_init_done = False

    def inverse(self, x: Tensor) -> tuple[Tensor, Tensor]:
        self._maybe_init(x)
        return super().inverse(x)

    def _maybe_init(self, x: Tensor) -> None:
        """The reference initialises from "the very first batch" (affine_constant_flow.py:42-50).  When that batch is
        sharded over the ranks of a ``torch.distributed`` group (rows shard, parameters are replicated),
        the statistics are those of the GLOBAL batch: three all-reduces of float64 column sums -- (count, sum x), then
        sum (x - mean)^2, then sum x e^s -- so that every rank ends up with the same ``s`` and ``t``, bit for bit, and
        the same values a single process would get from the concatenated batch (up to fp32 rounding of its own sums).
        Local statistics would make the replicas' parameters diverge silently."""
        if self.data_dep_init_done is False:
            with torch.no_grad():  # one-off, cross-row statistics: plain device ops
                sharded = _dist_world() > 1
                if not bool((self.s.squeeze() == 0).all()):
                    if sharded:
                        self.s.data = _dist.global_column_std(x).log().to(self.s.dtype).reshape(1, -1)
                    else:
                        self.s.data = x.std(dim=0, keepdim=True).log().detach()
                if not bool((self.t.squeeze() == 0).all()):
                    if sharded:
                        self.t.data = _dist.global_column_mean(x * self.s.exp()).to(self.t.dtype).reshape(1, -1)
                    else:
                        self.t.data = (x * self.s.exp()).mean(dim=0, keepdim=True).detach()
            self.data_dep_init_done = True

    def _run(self, x, inverse, accum):
        if inverse:
            self._maybe_init(x)
        return super()._run(x, inverse, accum)


_GLOW_WEIGHT_MAX_DIM = 64  # include/mnf_hip.h MNF_GLOW_WEIGHT_MAX_DIM


class _GlowWeightFn(torch.autograd.Function):
    """Glow's parameter preparation with gradients (mnf_glow_weight / _bwd): (W or W^-1, +-sum log|S|) from P, L, S, U in
    one launch each way -- the inverse by two triangular substitutions, no factorisation.  ``home``: see _RnvpFn."""

    @staticmethod
    def forward(ctx, L, S, U, P, inverse, home):
        Lc, Sc, Uc = (t.detach().contiguous() for t in (L, S, U))
        d = Sc.numel()
        out = torch.empty(d, d, dtype=torch.float32, device=Sc.device)
        ld = torch.empty((), dtype=torch.float32, device=Sc.device)
        _lib.check("mnf_glow_weight", _lib.load().mnf_glow_weight(
            P.data_ptr(), Lc.data_ptr(), Sc.data_ptr(), Uc.data_ptr(), out.data_ptr(), ld.data_ptr(), d, int(inverse),
            _stream()))
        ctx.save_for_backward(Lc, Sc, Uc, P, out)  # (out: in the inverse direction the gradient launch starts from W^-1)
        ctx.inverse, ctx.home = inverse, home
        ctx.set_materialize_grads(False)
        return out, ld

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out, g_ld):
        Lc, Sc, Uc, P, out = ctx.saved_tensors
        d, home = Sc.numel(), ctx.home
        go = None if g_out is None else g_out.contiguous()
        gl = None if g_ld is None else g_ld.contiguous()
        if home is not None:  # L, S, U back to back in a train.FlatParameters buffer: added to their gradient slice
            buf = home[0].grad[home[1]:home[1] + home[2]]
        else:
            buf = torch.empty(2 * d * d + d, dtype=torch.float32, device=Sc.device)
        gL, gS, gU = buf[:d * d], buf[d * d:d * d + d], buf[d * d + d:]
        _lib.check("mnf_glow_weight_bwd", _lib.load().mnf_glow_weight_bwd(
            P.data_ptr(), Lc.data_ptr(), Sc.data_ptr(), Uc.data_ptr(), _ptr(go), _ptr(gl), gL.data_ptr(), gS.data_ptr(),
            gU.data_ptr(), d, int(ctx.inverse), int(home is not None), out.data_ptr() if ctx.inverse else None, _stream()))
        if home is not None:
            return None, None, None, None, None, None
        return gL.view(d, d), gS, gU.view(d, d), None, None, None


class Glow(_TwoWayFlow):
    """Invertible d x d linear map, PLU-parametrised (flows/glow.py:5-37).

    ``P`` is a plain attribute (not in the state_dict), as in the reference.  W (and, for
    ``inverse``, its dense inverse) is d x d parameter preparation and is cached until a
    parameter changes -- the reference rebuilds both on every call (:20-24, :33-34); the
    row transform x @ W runs in the HIP library."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = int(dim)
        q, _ = torch.linalg.qr(torch.randn(dim, dim))
        P, L, U = torch.linalg.lu(q)
        self.P = P
        self.L = nn.Parameter(L)
        self.S = nn.Parameter(U.diag().clone())
        self.U = nn.Parameter(torch.triu(U, diagonal=1))
        self._w_key = None
        self._w: Tensor | None = None
        self._w_inv: Tensor | None = None
        self._w_ld: Tensor | None = None      # sum(log|S|), cached with W
        self._w_img: dict = {}                # inverse? -> MFMA operand image of W / W^-1
        self._w_index: Tensor | None = None

    def _P_on(self, device) -> Tensor:
        """``P`` on ``device`` (P is a plain attribute -- it does not follow ``.to()`` -- so it may live on the host:
        the copy is made once per (P, device), not per call, and never under a hipGraph capture)."""
        cached = self.__dict__.get("_P_dev")
        if cached is None or cached[0] is not self.P or cached[1].device != device:
            cached = (self.P, self.P.to(device))
            self.__dict__["_P_dev"] = cached
        return cached[1]

    def _assemble_W(self, device=None) -> Tensor:
        device = self.L.device if device is None else device
        L = torch.tril(self.L.detach(), diagonal=-1) + torch.eye(self.dim, device=self.L.device)
        U = torch.triu(self.U.detach(), diagonal=1)
        W = self._P_on(self.L.device) @ L @ (U + self.S.detach().diag())
        return W.to(device=device, dtype=torch.float32).contiguous()

    def _weights(self, device, inverse: bool) -> Tensor:
        key = (device, _flat_gen(self), tuple((p.data_ptr(), p._version) for p in (self.L, self.S, self.U)), id(self.P))
        if key != self._w_key:
            fused = self._fused_weight(device, False)
            if fused is not None:
                self._w, self._w_ld = fused
            else:
                self._w = self._assemble_W(device)
                self._w_ld = self.S.detach().abs().log().sum().to(device)  # 0-dim, parameter-only (glow.py:29)
            self._w_inv = None
            self._w_img = {}
            self._w_key = key
            self._w_from = (torch.cat([p.detach().reshape(-1) for p in (self.L, self.S, self.U)]).clone()
                            if _CHECK_PARAMS_EVERY else None)
        elif _CHECK_PARAMS_EVERY:
            _check_params_fresh((self.L, self.S, self.U), self.__dict__.get("_w_from"), "Glow")
        if inverse:
            if self._w_inv is None:
                fused = self._fused_weight(device, True)  # (two triangular substitutions: no LU, no host round trip)
                self._w_inv = fused[0] if fused is not None else torch.inverse(self._w).contiguous()
            return self._w_inv
        return self._w

    def _fused_weight(self, device, inverse: bool):
        """(W or W^-1, sum log|S|) from ``mnf_glow_weight``, or None when the shape / placement has no such kernel."""
        if (self.dim > _GLOW_WEIGHT_MAX_DIM or not self.L.is_cuda or self.L.device != device
                or self.L.dtype != torch.float32):
            return None
        out = torch.empty(self.dim, self.dim, dtype=torch.float32, device=device)
        ld = torch.empty((), dtype=torch.float32, device=device)
        P = self._P_on(device).to(torch.float32).contiguous()
        _lib.check("mnf_glow_weight", _lib.load().mnf_glow_weight(
            P.data_ptr(), self.L.detach().contiguous().data_ptr(), self.S.detach().contiguous().data_ptr(),
            self.U.detach().contiguous().data_ptr(), out.data_ptr(), ld.data_ptr(), self.dim, int(inverse), _stream()))
        return out, (-ld if inverse else ld)

    def _w_image(self, device, inverse: bool) -> Tensor | None:
        """W (or W^-1) in MFMA operand order, or None when dim has no specialised kernel."""
        W = self._weights(device, inverse)
        if inverse not in self._w_img:
            lib = _lib.load()
            n = lib.mnf_linear_rows_image_floats(self.dim)
            if n <= 0 or self.force_generic:
                self._w_img[inverse] = None
            else:
                if self._w_index is None or self._w_index.device != device:
                    idx = (ctypes.c_int32 * n)()
                    _lib.check("mnf_linear_rows_image_index", lib.mnf_linear_rows_image_index(self.dim, idx))
                    self._w_index = torch.frombuffer(idx, dtype=torch.int32).clone().to(device)
                img = torch.empty(n, dtype=torch.float32, device=device)
                _lib.check("mnf_pack_gather", lib.mnf_pack_gather(
                    W.data_ptr(), self._w_index.data_ptr(), img.data_ptr(), n, _stream()))
                self._w_img[inverse] = img
        return self._w_img[inverse]

    def _run(self, x, inverse, accum):
        if accum is None and isinstance(x, Tensor) and x.is_cuda and x.shape[0] > 0 and _wants_grad(self, x):
            xg = _grad_input(x)
            if (self.dim <= _GLOW_WEIGHT_MAX_DIM and self.L.is_cuda and self.L.device == xg.device
                    and self.L.dtype == torch.float32):
                # W (or W^-1, by triangular substitution) and log_det with their gradients: one launch each way
                params = [self.L, self.S, self.U]
                home = _flat_home_of(self, params) if all(p.requires_grad for p in params) else None
                M, ld = _GlowWeightFn.apply(self.L, self.S, self.U, self._P_on(xg.device).to(torch.float32).contiguous(),
                                            bool(inverse), home)
                return _LinearRowsFn.apply(xg, M), ld
            eye = torch.eye(self.dim, device=self.L.device)
            W = self._P_on(self.L.device) @ (torch.tril(self.L, diagonal=-1) + eye) @ (
                torch.triu(self.U, diagonal=1) + self.S.diag())  # glow.py:20-24, differentiable
            ld = self.S.abs().log().sum()
            if inverse:
                # (inv_ex: torch.inverse's singularity check is a host synchronisation, which a hipGraph capture of the
                #  training step cannot record; a singular W shows up as inf/nan in the loss either way)
                return _LinearRowsFn.apply(xg, torch.linalg.inv_ex(W).inverse.to(xg.device)), -ld.to(xg.device)
            return _LinearRowsFn.apply(xg, W.to(xg.device)), ld.to(xg.device)
        x = _device_input(x, "input")
        if x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        W = self._weights(x.device, inverse)
        img = self._w_image(x.device, inverse)
        y = torch.empty_like(x)
        if x.shape[0] > 0:
            if img is not None:
                _lib.check("mnf_linear_rows_img", _lib.load().mnf_linear_rows_img(
                    x.data_ptr(), img.data_ptr(), y.data_ptr(), x.shape[0], self.dim, _stream()))
            else:
                _lib.check("mnf_linear_rows", _lib.load().mnf_linear_rows(
                    x.data_ptr(), W.data_ptr(), y.data_ptr(), x.shape[0], self.dim, _stream()))
        ld = -self._w_ld if inverse else self._w_ld
        if accum is not None:
            accum += ld
            return y, None
        return y, ld


class _SplineBlockRun:
    """One ``[ActNormFlow, Glow, NSF_CL]`` block (readme.md:67-73) as one ``mnf_nsf_cl_fused`` launch: ActNorm
    and Glow are folded into one ``row @ A + b`` that the spline kernel applies to the rows while they are in
    registers.  A plain object (not a Module): it caches the folded operand image per direction."""

    def __init__(self, actnorm: "ActNormFlow", glow: "Glow", nsf: "NSF_CL") -> None:
        self.actnorm, self.glow, self.nsf = actnorm, glow, nsf
        self.dim = nsf.dim
        self._aff_key = None
        self._aff: dict = {}
        self._lin_index: Tensor | None = None
        self._unsupported = False

    def _affine(self, device, inverse: bool):
        """(aff buffer = [operand image of A | b], log-det constant, [exp(s) | t]) for one direction, cached."""
        an, gl = self.actnorm, self.glow
        key = (device, _flat_gen(an), tuple((p.data_ptr(), p._version) for p in (an.s, an.t, gl.L, gl.S, gl.U)), id(gl.P))
        if key != self._aff_key:
            self._aff, self._aff_key = {}, key
            self._aff_from = (torch.cat([p.detach().reshape(-1) for p in (an.s, an.t, gl.L, gl.S, gl.U)]).clone()
                              if _CHECK_PARAMS_EVERY else None)
        elif _CHECK_PARAMS_EVERY:
            _check_params_fresh((an.s, an.t, gl.L, gl.S, gl.U), self.__dict__.get("_aff_from"), "an [ActNorm, Glow] block")
        if inverse not in self._aff:
            lib = _lib.load()
            n = lib.mnf_linear_rows_image_floats(self.dim)
            if n <= 0:
                self._aff[inverse] = None
                return None
            s = an.s.detach().to(device, torch.float32).reshape(-1)
            t = an.t.detach().to(device, torch.float32).reshape(-1)
            W = gl._weights(device, inverse)  # W (forward) or W^-1 (inverse), cached by Glow
            if inverse:   # z = (y @ W^-1 - t) e^-s
                A, b = W * torch.exp(-s)[None, :], -t * torch.exp(-s)
                ldc = -(s.sum() + gl._w_ld)
            else:         # x = (z e^s + t) @ W
                A, b = torch.exp(s)[:, None] * W, t @ W
                ldc = s.sum() + gl._w_ld
            if self._lin_index is None or self._lin_index.device != device:
                idx = (ctypes.c_int32 * n)()
                _lib.check("mnf_linear_rows_image_index", lib.mnf_linear_rows_image_index(self.dim, idx))
                self._lin_index = torch.frombuffer(idx, dtype=torch.int32).clone().to(device)
            aff = torch.empty(n + self.dim, dtype=torch.float32, device=device)
            _lib.check("mnf_pack_gather", lib.mnf_pack_gather(
                A.contiguous().data_ptr(), self._lin_index.data_ptr(), aff.data_ptr(), n, _stream()))
            aff[n:] = b
            scale_shift = torch.cat((torch.exp(s), t)).contiguous()
            self._aff[inverse] = (aff, float(ldc), scale_shift)  # one host read per parameter update
        return self._aff[inverse]

    def usable(self, x, inverse: bool) -> bool:
        needs_init = inverse and self.actnorm.data_dep_init_done is False
        return (not self._unsupported and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0
                and x.shape[1] == self.dim and not needs_init and not self.nsf.force_generic
                and not self.glow.force_generic
                and not any(_wants_grad(m, x) for m in (self.actnorm, self.glow, self.nsf)))

    def launch(self, x: Tensor, inverse: bool, log_det: Tensor, accumulate: bool, keep: bool,
               logprob: tuple | None = None) -> list[Tensor] | None:
        """Returns the output tensors in application order -- all three when ``keep`` (each written once from
        registers, never re-read), else just the last -- or None when the shape has no fused kernel.
        ``logprob`` = (log_prob (rows,), zeroed fp64 sum (1,) or None): the kernel also does the standard-normal
        log-prob epilogue (the block is the last launch of a density pass)."""
        nsf = self.nsf
        packed = self._affine(x.device, inverse)
        _, image = nsf._packed(x.device)
        if packed is None or image is None:
            self._unsupported = True
            return None
        aff, ldc, scale_shift = packed
        x = _device_input(x, "input")
        buf = torch.empty((3 if keep else 1, x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        rc = _lib.load().mnf_nsf_cl_fused(
            x.data_ptr(), buf[-1].data_ptr(), log_det.data_ptr(), int(accumulate), image.data_ptr(),
            _ptr(nsf._split_image(x.device)), aff.data_ptr(), ldc, scale_shift.data_ptr(),
            buf[0].data_ptr() if keep else None, buf[1].data_ptr() if keep else None,
            _ptr(logprob[0]) if logprob else None, _ptr(logprob[1]) if logprob else None, x.shape[0], self.dim, nsf.K, float(nsf.B), int(inverse), len(nsf.h_sizes), nsf._hid, _stream())
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            # (an fp32 request has no fused kernel at any shape; it is a per-call choice, not a property of the shape)
            self._unsupported = not nsf._fp32_request()
            return None
        _lib.check("mnf_nsf_cl_fused", rc)
        return list(buf.unbind(0))


class FusedSplineBlock(_TwoWayFlow):
    """Opt-in fusion of the reference's ``[ActNormFlow, Glow, NSF_CL]`` block (readme.md:67-73) into one
    kernel launch per direction that writes NO intermediate tensor.

    The three sub-modules keep their parameters (``state_dict`` keys ``actnorm.*``, ``glow.*``,
    ``nsf.*``).  As a member of a ``NormalizingFlow`` the block contributes one entry to the returned list
    instead of three, which is why this is an explicit opt-in (``NormalizingFlow`` itself runs such a block
    as one launch too, but still writes both intermediates).  Falls back to running the three modules in
    sequence when a graph is being recorded, for ActNorm's data-dependent first ``inverse`` call, and for
    shapes without a fused kernel."""

    def __init__(self, actnorm: ActNormFlow, glow: Glow, nsf: NSF_CL) -> None:
        super().__init__()
        if not (actnorm.dim == glow.dim == nsf.dim):
            raise ValueError("the three layers must share dim")
        self.actnorm, self.glow, self.nsf = actnorm, glow, nsf
        self.dim = nsf.dim
        self.__dict__["_run_helper"] = _SplineBlockRun(actnorm, glow, nsf)

    def _packed_params(self) -> list[Tensor]:
        return []

    def invalidate(self) -> None:
        super().invalidate()
        self._run_helper._aff_key = None

    def _sequence(self, x: Tensor, inverse: bool):
        order = (self.nsf, self.glow, self.actnorm) if inverse else (self.actnorm, self.glow, self.nsf)
        ld = 0
        for m in order:
            x, l1 = m._run(x, inverse, None)
            ld = ld + l1
        return x, ld

    def _run(self, x, inverse, accum):
        run = self._run_helper
        if run.usable(x, inverse):
            ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
            out = run.launch(x, inverse, ld, accum is not None, keep=False)
            if out is not None:
                return out[-1], (None if accum is not None else ld)
        elif isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        y, ld = self._sequence(x, inverse)
        if accum is not None:
            accum += ld
            return y, None
        return y, ld


class _AffineRun:
    """Consecutive ``AffineHalfFlow`` layers of one shape, in MODEL order, as one ``mnf_affine_half_stack``
    launch.  A plain object (not a Module): it only caches the concatenated operand images."""

    MAX_LAYERS = 32  # mnf_affine_half_stack takes the layers' parities as one 32-bit word

    def __init__(self, layers: Sequence["AffineHalfFlow"]) -> None:
        self.layers = list(layers)
        if len(self.layers) > self.MAX_LAYERS:
            raise ValueError(f"one stack launch covers at most {self.MAX_LAYERS} layers")
        self._key = None
        self._images: Tensor | None = None
        self._splits: Tensor | None = None
        self._plist = None
        self._unsupported = False  # set once the library reports that the shape has no stack kernel
        self._no_fused_logprob = False  # set once the library reports that the shape has no fused log-prob epilogue
        self.logprob_fused = False
        self._flat_home = None   # (FlatParameters, offset, length) while the run's parameters live in one buffer
        self._flat_checked = None
        self._grad_i0 = None     # index of the run's first parameter in FlatParameters.params
        self._home_now = None    # the flat home of the call in flight (set by launch_grad, read by _AffineRunFn)
        self._stand_in = None

    @staticmethod
    def compatible(a: "AffineHalfFlow", b: "AffineHalfFlow") -> bool:
        return (a.dim, a.h_sizes, a.scale, a.shift) == (b.dim, b.h_sizes, b.scale, b.shift)

    def _params(self) -> list[Tensor]:
        """Every layer's packed parameters in order; the walk is redone only when a conditioner (or one of its
        Linear children) has been replaced."""
        nets = [n for f in self.layers for n in (f._modules.get("s_net"), f._modules.get("t_net")) if n is not None]
        sig = [id(c) for n in nets for c in n._modules.values()] + [id(n) for n in nets]
        if self._plist is None or self._plist[0] != sig:
            self._plist = (sig, [p for f in self.layers for p in f._packed_params()])
        return self._plist[1]

    def flat_home(self):
        """(FlatParameters, offset, length) if every parameter of the run is a view of one train.FlatParameters
        buffer, back to back in order -- then the kernels use that slice (no concatenation, no gradient scatter) --
        else None.  Re-checked only when the FlatParameters object changes."""
        flat = self.layers[0].__dict__.get("_mnf_flat")
        if flat is not self._flat_checked:
            self._flat_checked, self._flat_home = flat, None
            if flat is not None:
                sl = flat.slice_of(self._params())
                if sl is not None:
                    self._flat_home = (flat, sl[0], sl[1])
        home = self._flat_home
        if home is not None:
            # Re-validated on every call (two data pointers, the requires_grad flags, one gradient view): after
            # model.to() / .float() / load_state_dict(assign=True) the parameters no longer live in the buffer, a
            # parameter frozen with requires_grad_(False) must not receive the in-place gradient sums, and a torch
            # optimizer's zero_grad(set_to_none=True) detaches p.grad from the gradient buffer (the in-place sums
            # would then be invisible to it).  In each case the run takes the ordinary path: one concatenation,
            # gradients returned to autograd.  (torch.autograd.grad() on a flat-homed run still ADDS into
            # FlatParameters.grad as a side effect and returns no parameter gradients: use .backward() there.)
            if not self._home_ok(home[0], self._params()):
                return None
        return home

    def _home_ok(self, flat, params) -> bool:
        if not flat.home_is_valid(params):
            return False
        i0 = self._grad_i0
        if i0 is None or flat.params[i0] is not params[0]:
            try:
                i0 = self._grad_i0 = next(i for i, q in enumerate(flat.params) if q is params[0])
            except StopIteration:
                return False
        return params[0].grad is flat._grad_views[i0] and params[-1].grad is flat._grad_views[i0 + len(params) - 1]

    def images(self, device, flat: Tensor | None = None):
        """(fp32 operand images, split operand images) of all layers, back to back; repacked after a weight update
        with ONE parameter concatenation (``flat``: the caller's, if it already has one) and one launch per kind."""
        params = self._params()
        home = self.flat_home()
        if home is not None:  # one counter instead of 16 (data_ptr, version) pairs per layer
            key = (device, [f.force_fp32_mfma or not f._split_ok for f in self.layers], id(home[0]), home[0].generation)
            if flat is None:
                flat = home[0].data[home[1]:home[1] + home[2]]
        else:
            key = (device, [f.force_fp32_mfma or not f._split_ok for f in self.layers],
                   [(p.data_ptr(), p._version) for p in params])
        if key != self._key:
            f0, n = self.layers[0], len(self.layers)
            index = f0._device_index(device)
            self._images = self._splits = None
            if index is None:  # the shape has no specialised kernel at all: a static property
                self._unsupported = True
            else:
                lib = _lib.load()
                if flat is None:
                    flat = torch.cat([p.detach().reshape(-1) for p in params])
                flat = flat.to(device=device, dtype=torch.float32).contiguous()
                stride = flat.numel() // n
                self._images = torch.empty(n * index.numel(), dtype=torch.float32, device=device)
                _lib.check("mnf_pack_gather_batch", lib.mnf_pack_gather_batch(
                    flat.data_ptr(), index.data_ptr(), self._images.data_ptr(), index.numel(), n, stride, _stream()))
                sidx = f0._device_split_index(device)
                if sidx and not _FP32_MFMA_ENV and not any(f.force_fp32_mfma or not f._split_ok for f in self.layers):
                    table, n_split, n_plain = sidx
                    self._splits = torch.empty(n * (n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS), dtype=torch.int32,
                                               device=device)
                    _lib.check("mnf_pack_gather_split_batch", lib.mnf_pack_gather_split_batch(
                        flat.data_ptr(), table.data_ptr(), self._splits.data_ptr(), n_split, n_plain, n, stride,
                        _stream()))
                # (debug switch MNF_CHECK_PARAMS: keep what the images were packed from -- with a flat home `flat` is
                #  the live buffer itself, so a copy)
                self._packed_from = flat.detach().clone() if _CHECK_PARAMS_EVERY else None
            self._key = key
        elif _CHECK_PARAMS_EVERY:
            _check_params_fresh(params, self.__dict__.get("_packed_from"), "a run of AffineHalfFlow layers")
        return self._images, self._splits

    def bwd_images(self, device, flat: Tensor):
        """(split gradient-kernel images of all layers back to back, words per image) for the parameters in ``flat``
        (the run's concatenation, model order), or None when the gradients run on the fp32 kernels.  Packed per
        backward pass with one launch: the weights change between steps."""
        f0, n = self.layers[0], len(self.layers)
        if not all(f._bwd_split_ok() for f in self.layers):
            return None
        table = f0._bwd_split_index(device)
        if not table:
            return None
        idx, n_split, n_plain = table
        words = n_split + n_plain + _lib.MNF_SPLIT_TAIL_WORDS
        images = torch.empty(n * words, dtype=torch.int32, device=device)
        _lib.check("mnf_pack_gather_split_batch", _lib.load().mnf_pack_gather_split_batch(
            flat.data_ptr(), idx.data_ptr(), images.data_ptr(), n_split, n_plain, n, flat.numel() // n, _stream()))
        return images, words

    def ready(self, x):
        """The run's (fp32 images, split images) if ``x`` can go through the stack kernel without gradients, else
        None.  Hand the result to ``launch(images=...)``: the cache is then validated once per pass, not twice."""
        if (self._unsupported or not isinstance(x, Tensor) or not x.is_cuda or x.dim() != 2 or x.shape[0] == 0
                or x.shape[1] != self.layers[0].dim or any(f.force_generic for f in self.layers)
                or any(_wants_grad(f, x) for f in self.layers)):
            return None
        imgs = self.images(x.device)
        return imgs if imgs[0] is not None else None

    def usable(self, x) -> bool:
        return self.ready(x) is not None

    def trainable(self, x) -> bool:
        """Gradients wanted and the whole run can go through one autograd node (stack kernel forward; backward
        layer by layer on the saved intermediates: the fp32-MFMA gradient kernel, or the generic one for shapes
        it does not cover)."""
        return (not self._unsupported and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0
                and x.shape[1] == self.layers[0].dim and not any(f.force_generic for f in self.layers)
                and any(_wants_grad(f, x) for f in self.layers)
                and self.layers[0]._device_index(x.device) is not None)

    def launch_grad(self, x: Tensor, inverse: bool, with_lp: bool = False):
        """(outputs in application order, log_det) with the autograd link; None when the shape has no kernels.
        ``with_lp``: log p under a standard-normal base instead, as ONE tensor (see _AffineRunFn.forward)."""
        # (a refusal of the fused log-prob form is remembered per CALL shape: it can depend on the row count and on the
        #  batch's alignment, not only on the layer -- one misaligned batch must not send every later call the slow way)
        lp_key = (x.shape[0], x.shape[1], x.device, x.data_ptr() % 16)
        if with_lp and lp_key in getattr(self, "_lp_unsupported", ()):
            return None
        self._home_now = self.flat_home()
        if self._home_now is not None:
            if self._stand_in is None or self._stand_in.device != x.device:
                self._stand_in = torch.zeros(1, device=x.device, requires_grad=True)
            flat = self._stand_in
        else:
            flat = torch.cat([p.reshape(-1) for p in self._params()])
        try:
            out = _AffineRunFn.apply(_grad_input(x), flat, self, bool(inverse), bool(with_lp))
        except MnfHipError as err:  # an image exists but no stack kernel (e.g. hidden width 32): layer by layer
            if err.code != _lib.MNF_ERR_UNSUPPORTED:
                raise
            if with_lp:  # (only the epilogue may be missing: the caller takes the unfused route -- and does not ask
                refused = set(getattr(self, "_lp_unsupported", ()))  # again: the refusal comes AFTER a whole stack launch)
                if len(refused) > 64:
                    refused.clear()
                refused.add(lp_key)
                self._lp_unsupported = refused
                return None
            self._unsupported = True
            return None
        if with_lp:
            return out
        return list(out[:-1]), out[-1]

    def launch(self, x: Tensor, inverse: bool, log_det: Tensor, accumulate: bool, sqnorm: Tensor | None,
               keep: bool, logprob: tuple | None = None, images: tuple | None = None) -> list[Tensor] | None:
        """Runs the layers (model order reversed when ``inverse``).  Returns the output tensors in
        application order -- all of them when ``keep`` (views of one buffer: each intermediate is written
        once and never re-read), else just the last -- or None when the shape has no stack kernel.

        ``logprob`` = (log_prob (rows,), fp64 sum (1,) zeroed, or None): ask the kernel for the
        standard-normal log-prob epilogue too; ``self.logprob_fused`` says whether it did (only the split
        kernel can -- otherwise ``sqnorm`` is filled as usual and the caller runs the epilogue kernel)."""
        f0, n = self.layers[0], len(self.layers)
        images, splits = images if images is not None else self.images(x.device)  # (``images``: from ready())
        x = _device_input(x, "input")
        buf = torch.empty((n if keep else 1, x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        par = _lib.int_array([int(bool(f.parity)) for f in self.layers])

        def go(lp, total, sq):
            return _lib.load().mnf_affine_half_stack(
                x.data_ptr(), buf[-1].data_ptr(), buf.data_ptr() if keep and n > 1 else None, log_det.data_ptr(),
                _ptr(sq), _ptr(lp), _ptr(total), int(accumulate), images.data_ptr(), _ptr(splits), par, n, x.shape[0],
                f0.dim, int(inverse), len(f0.h_sizes), f0._hid, _stream())

        self.logprob_fused = (logprob is not None and splits is not None and not self._no_fused_logprob
                              and not _dispatch.NO_FUSED_LOGPROB)
        rc = go(logprob[0], logprob[1], None) if self.logprob_fused else go(None, None, sqnorm)
        if rc == _lib.MNF_ERR_UNSUPPORTED and self.logprob_fused:  # shape runs on the fp32 stack kernel: no epilogue
            self._no_fused_logprob, self.logprob_fused = True, False
            rc = go(None, None, sqnorm)
        if rc == _lib.MNF_ERR_UNSUPPORTED:
            self._unsupported = True
            return None
        _lib.check("mnf_affine_half_stack", rc)
        return list(buf.unbind(0))


class FusedAffineStack(_TwoWayFlow):
    """Opt-in whole-stack fusion of consecutive ``AffineHalfFlow`` layers of one shape (SURVEY.md 8f
    rank 3): all of them in ONE kernel launch per direction, the rows held in registers across
    layers -- HBM sees each row once in and once out.  No intermediate tensor exists, so as a member
    of a ``NormalizingFlow`` it contributes ONE entry to the returned list instead of one per layer;
    that change of what ``forward``/``inverse`` return is why this is an explicit opt-in.
    (``NormalizingFlow`` itself fuses runs of equal AffineHalfFlow layers while KEEPING every
    intermediate; this class is for callers that do not want them.)

    ``state_dict`` keys are ``layers.{i}.s_net...``; falls back to running the layers one by one when
    a graph is being recorded or the shape has no fused kernel."""

    def __init__(self, layers: Sequence[AffineHalfFlow]) -> None:
        super().__init__()
        layers = list(layers)
        if not layers or any(not isinstance(f, AffineHalfFlow) for f in layers):
            raise TypeError("FusedAffineStack takes AffineHalfFlow layers")
        f0 = layers[0]
        if any(not _AffineRun.compatible(f, f0) for f in layers):
            raise ValueError("all layers must share dim, h_sizes and the scale/shift flags")
        self.layers = nn.ModuleList(layers)
        self.dim = f0.dim
        # not Modules: keep them out of the module tree.  One launch per chunk of at most 32 layers.
        m = _AffineRun.MAX_LAYERS
        self.__dict__["_run_helpers"] = [_AffineRun(layers[k:k + m]) for k in range(0, len(layers), m)]

    def _packed_params(self) -> list[Tensor]:
        return []

    def invalidate(self) -> None:
        super().invalidate()
        for r in self._run_helpers:
            r._key = None

    def _sequence(self, x, inverse, sqnorm=None):
        ld = 0
        order = list(reversed(self.layers)) if inverse else list(self.layers)
        for i, f in enumerate(order):
            x, l1 = f._run(x, inverse, None, sqnorm if i == len(order) - 1 else None)
            ld = ld + l1
        return x, ld

    def emits_sqnorm(self, device) -> bool:
        return all(r.images(device)[0] is not None for r in self._run_helpers)

    def _run(self, x, inverse, accum, sqnorm: Tensor | None = None, overwrite: bool = False):
        runs = list(reversed(self._run_helpers)) if inverse else list(self._run_helpers)
        imgs = [r.ready(x) for r in runs]
        if all(i is not None for i in imgs):
            ld = accum if accum is not None else torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
            y, done = x, 0
            for k, (run, im) in enumerate(zip(runs, imgs)):
                out = run.launch(y, inverse, ld, k > 0 or (accum is not None and not overwrite),
                                 sqnorm if k == len(runs) - 1 else None, keep=False, images=im)
                if out is None:
                    break
                y, done = out[-1], k + 1
            if done == len(runs):
                return y, (None if accum is not None else ld)
            if done > 0:
                # a later chunk has no stack kernel AFTER earlier chunks have already added their log-dets into ld:
                # go on from y layer by layer (restarting from x would count the first chunks twice)
                rest = [f for r in runs[done:] for f in (reversed(r.layers) if inverse else r.layers)]
                for i, f in enumerate(rest):
                    y, _ = f._run(y, inverse, ld, sqnorm if i == len(rest) - 1 else None)
                return y, (None if accum is not None else ld)
        elif isinstance(x, Tensor) and x.dim() == 2 and x.shape[1] != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {x.shape[1]}")
        y, ld = self._sequence(x, inverse, sqnorm)
        if accum is not None:
            if overwrite:
                accum.copy_(ld)
            else:
                accum += ld
            return y, None
        return y, ld


class _GlowActNormInvLogProbFn(torch.autograd.Function):
    """The pair closing a density pass under a standard-normal base: log p from (u, the running log_det) in one launch --
    z is not written --, and one gradient launch that forms grad_z = -z d loss / d log p from the recomputed z
    (mnf_glow_actnorm_inv_logprob / _bwd).  Replaces the pair node + gauss_logprob + the -z g elementwise launch."""

    @staticmethod
    def forward(ctx, u, M, s, t, ld_glow, log_det):
        Mc = M.detach().contiguous()
        sc = s.detach().to(u.device, torch.float32).contiguous()
        tc = t.detach().to(u.device, torch.float32).contiguous()
        lp = torch.empty(u.shape[0], dtype=torch.float32, device=u.device)
        _lib.check("mnf_glow_actnorm_inv_logprob", _lib.load().mnf_glow_actnorm_inv_logprob(
            u.data_ptr(), Mc.data_ptr(), sc.data_ptr(), tc.data_ptr(), ld_glow.detach().data_ptr(),
            log_det.detach().contiguous().data_ptr(), lp.data_ptr(), u.shape[0], u.shape[1], _stream()))
        ctx.save_for_backward(u, Mc, sc, tc)
        ctx.ld_shape = ld_glow.shape
        return lp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_lp):
        u, Mc, sc, tc = ctx.saved_tensors
        dim = u.shape[1]
        g = grad_lp.contiguous()
        gu = torch.empty_like(u)
        sums = torch.zeros(dim * dim + 2 * dim + 1, dtype=torch.float32, device=u.device)  # grad_M | grad_s | grad_t | grad_ld
        gM, gs = sums[:dim * dim], sums[dim * dim:dim * dim + dim]
        gt, gld = sums[dim * dim + dim:dim * dim + 2 * dim], sums[dim * dim + 2 * dim:]
        work = _pair_bwd_workspace(u.shape[0], dim, u.device)
        _lib.check("mnf_glow_actnorm_inv_logprob_bwd", _lib.load().mnf_glow_actnorm_inv_logprob_bwd_det(
            u.data_ptr(), g.data_ptr(), Mc.data_ptr(), sc.data_ptr(), tc.data_ptr(), gu.data_ptr(), gM.data_ptr(),
            gs.data_ptr(), gt.data_ptr(), gld.data_ptr(), u.shape[0], dim, _ptr(work), 0 if work is None else work.numel(),
            _stream()))
        return (gu if ctx.needs_input_grad[0] else None, gM.view(dim, dim), gs.view(sc.shape), gt.view(tc.shape),
                gld.reshape(ctx.ld_shape), g)


def _pair_fusable(glow: "Glow", actnorm: "ActNormFlow", x) -> bool:
    """Training pass, x -> z: can Glow.inverse + ActNormFlow.inverse at this input go out as the fused pair?"""
    return (not _NO_PAIR_FUSION_ENV and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0
            and x.dtype == torch.float32 and glow.dim == actnorm.dim == x.shape[1] and glow.dim in _PAIR_FUSION_DIMS
            and actnorm.data_dep_init_done is not False and not glow.force_generic
            and glow.L.is_cuda and glow.L.device == x.device and glow.L.dtype == torch.float32
            and _wants_grad(glow, x) and _wants_grad(actnorm, x))


def _glow_actnorm_inverse(glow: "Glow", actnorm: "ActNormFlow", x: Tensor, log_det: Tensor | None = None):
    """(z, the pair's log|det J|); with ``log_det`` (rows,): log p under a standard-normal base instead (one tensor)."""
    xg = _grad_input(x)
    params = [glow.L, glow.S, glow.U]
    home = _flat_home_of(glow, params) if all(p.requires_grad for p in params) else None
    M, ld_glow = _GlowWeightFn.apply(glow.L, glow.S, glow.U, glow._P_on(xg.device).to(torch.float32).contiguous(), True,
                                     home)
    if log_det is not None:
        return _GlowActNormInvLogProbFn.apply(xg, M, actnorm.s.to(xg.device), actnorm.t.to(xg.device), ld_glow, log_det)
    return _GlowActNormInvFn.apply(xg, M, actnorm.s.to(xg.device), actnorm.t.to(xg.device), ld_glow)


class NormalizingFlow(nn.Module):
    """Runs flows in order (forward) or reversed (inverse), summing log|det J| and keeping every
    intermediate (flows/core.py:10-35).  Layers from this package accumulate ``log_det`` inside
    their kernel; any other duck-typed flow is called by name and added like the reference does."""

    def __init__(self, flows: Sequence[nn.Module]) -> None:
        super().__init__()
        self.flows = nn.ModuleList(flows)
        # bench.py: set to a list to collect a (start, end) HIP event pair per layer, recorded on
        # the stream the kernels are launched on (consecutive layers share the boundary event)
        self.layer_events: list | None = None
        self.layer_event_pick: int | None = None
        # runs of equal AffineHalfFlow layers go out as ONE launch that still writes every intermediate
        self.fuse_affine_runs = True
        self._last_sqnorm: Tensor | None = None

    def invalidate(self) -> None:
        """Drop every layer's packed operand images and the fused runs' (see ``_HipFlow.invalidate``)."""
        for f in self.modules():
            if f is not self and hasattr(f, "invalidate"):
                f.invalidate()
        self.__dict__.pop("_runs_cache", None)

    def _affine_runs(self) -> dict:
        """start index (model order) -> run object for every group of layers that goes out as one launch:
        maximal runs of >= 2 consecutive AffineHalfFlow layers of one shape (_AffineRun) and
        [ActNormFlow, Glow, NSF_CL] blocks (_SplineBlockRun); rebuilt when the module list changes."""
        ids = tuple(id(f) for f in self.flows)
        cache = self.__dict__.get("_runs_cache")
        if cache is None or cache[0] != ids:
            runs, flows, i = {}, list(self.flows), 0
            while i < len(flows):
                j = i + 1
                if type(flows[i]) is AffineHalfFlow:
                    while j < len(flows) and type(flows[j]) is AffineHalfFlow and _AffineRun.compatible(flows[i], flows[j]):
                        j += 1
                    # (one launch covers at most _AffineRun.MAX_LAYERS layers: longer runs go out in chunks)
                    for k in range(i, j, _AffineRun.MAX_LAYERS):
                        if min(j, k + _AffineRun.MAX_LAYERS) - k >= 2:
                            runs[k] = _AffineRun(flows[k:min(j, k + _AffineRun.MAX_LAYERS)])
                elif (i + 2 < len(flows) and type(flows[i]) is ActNormFlow and type(flows[i + 1]) is Glow
                      and type(flows[i + 2]) is NSF_CL and flows[i].dim == flows[i + 1].dim == flows[i + 2].dim):
                    runs[i] = _SplineBlockRun(flows[i], flows[i + 1], flows[i + 2])
                    j = i + 3
                i = j
            cache = (ids, runs)
            self.__dict__["_runs_cache"] = cache
        return cache[1]

    def _pass(self, x: Tensor, inverse: bool, want_sqnorm: bool = False, prologue=None, want_logprob=None,
              last_only: bool = False, lp_tail: bool = False):
        """prologue (forward only, first flow an RNVP): see RNVP._run; the returned list then starts with eps.
        last_only: the caller reads only the last tensor of the returned list (log_prob): pairs of layers with a fused
        training kernel go out as one autograd node and their intermediate is not materialised.
        lp_tail (with last_only, standard-normal base): when such a pair CLOSES the pass its node produces log p itself
        (``self._lp_node``; the list then ends with None and log_det is not extended).
        want_logprob = (lp, total-or-None): when the LAST launch is an affine run whose kernel can add the
        standard-normal epilogue, it fills them and ``self._logprob_done`` is set."""
        n = len(self.flows)
        order = list(reversed(self.flows)) if inverse else list(self.flows)
        runs = self._affine_runs() if self.fuse_affine_runs and not _NO_RUN_FUSION_ENV else {}
        # position in `order` -> run that starts there
        span_of = lambda r: len(r.layers) if isinstance(r, _AffineRun) else 3
        run_at = {(n - (start + span_of(r)) if inverse else start): r for start, r in runs.items()}
        # a first layer whose kernel writes log_det for every row saves zero-filling it
        fresh = (bool(order) and isinstance(order[0], (AffineHalfFlow, FusedAffineStack)) and isinstance(x, Tensor)
                 and x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and not _wants_grad(order[0], x))
        # a first layer on the autograd path returns its own per-row log_det: it BECOMES the running sum (no zero fill,
        # no add -- two launches of a step that is made of launches)
        lazy = (not fresh and bool(order) and isinstance(order[0], (RNVP, AffineHalfFlow, NSF_CL, NSF_AR, MAF))
                and (not inverse or isinstance(order[0], _TwoWayFlow)) and run_at.get(0) is None and prologue is None
                and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and _wants_grad(order[0], x))
        log_det = None if lazy else (torch.empty(x.size(0), device=x.device) if fresh
                                     else torch.zeros(x.size(0), device=x.device))
        seen = [x]
        self._last_sqnorm = None
        self._logprob_done = False
        self._lp_node = None
        # layer_events: list receiving (start, end, index) HIP events per launch; layer_event_pick = i restricts
        # the marks to the launch at position i of this pass (a mark costs a few us of stream time)
        pick = self.layer_event_pick
        events_on = self.layer_events is not None and x.is_cuda
        i = 0
        while i < n:
            flow = order[i]
            run = run_at.get(i)
            train_run = run is not None and isinstance(run, _AffineRun) and run.trainable(x)
            run_images = None  # an affine run's validated operand images, handed on to its launch
            if run is not None and not train_run:
                if isinstance(run, _AffineRun):
                    run_images = run.ready(x)
                    if run_images is None:
                        run = None
                elif not run.usable(x, inverse):
                    run = None
            span = span_of(run) if run is not None else 1
            last = i + span == n
            timed = events_on and (pick is None or pick == i)
            if timed:
                e_prev = torch.cuda.Event(enable_timing=True)
                e_prev.record()
            outs = None
            if run is not None:
                # one launch for the whole run; every intermediate is written once and never re-read
                if train_run:  # one autograd node for the whole run
                    sq = None
                    res = run.launch_grad(x, inverse)
                    if res is not None:
                        outs, ld_run = res
                        log_det = ld_run if (fresh and i == 0) else log_det + ld_run
                elif isinstance(run, _AffineRun):
                    sq = torch.empty(x.size(0), device=x.device) if (want_sqnorm and last) else None
                    outs = run.launch(x, inverse, log_det, not (fresh and i == 0), sq, keep=True,
                                      logprob=want_logprob if last else None, images=run_images)
                    if outs is not None and last and want_logprob is not None and run.logprob_fused:
                        self._logprob_done, sq = True, None
                else:
                    sq = None
                    outs = run.launch(x, inverse, log_det, True, keep=True,
                                      logprob=want_logprob if (last and not _dispatch.NO_FUSED_LOGPROB) else None)
                    if outs is not None and last and want_logprob is not None and not _dispatch.NO_FUSED_LOGPROB:
                        self._logprob_done = True
                if outs is not None:
                    self._last_sqnorm = sq
                    seen.extend(outs)
                    x = outs[-1]
                else:
                    span, last = 1, i + 1 == n
            if (outs is None and last_only and inverse and i + 1 < n and type(flow) is Glow
                    and type(order[i + 1]) is ActNormFlow and _pair_fusable(flow, order[i + 1], x)):
                # Glow.inverse + ActNormFlow.inverse of a training pass: one launch each way (csrc/mnf_glow_actnorm.hip)
                if (lp_tail and i + 2 == n and not _dispatch.NO_FUSED_LOGPROB and isinstance(log_det, Tensor)
                        and log_det.shape == (x.shape[0],) and log_det.dtype == torch.float32):
                    self._lp_node = _glow_actnorm_inverse(flow, order[i + 1], x, log_det)
                    x = None
                else:
                    x, ld = _glow_actnorm_inverse(flow, order[i + 1], x)
                    log_det = ld if log_det is None else log_det + ld
                seen.append(x)
                span, outs = 2, [x]
            if outs is None:
                if (want_sqnorm and last and isinstance(flow, (AffineHalfFlow, FusedAffineStack))
                        and x.is_cuda and x.shape[0] > 0 and not _wants_grad(flow, x) and flow.emits_sqnorm(x.device)):
                    # last layer also emits |z|^2 per row for the standard-normal epilogue
                    self._last_sqnorm = torch.empty(x.size(0), device=x.device)
                    x, _ = flow._run(x, inverse, log_det, self._last_sqnorm, overwrite=fresh and i == 0)
                elif isinstance(flow, _HipFlow) and (inverse is False or isinstance(flow, _TwoWayFlow)) \
                        and _wants_grad(flow, x):
                    x, ld = flow._run(x, inverse, None)    # autograd path: gradients from the *_bwd kernels
                    log_det = ld if log_det is None else log_det + ld
                elif prologue is not None and i == 0:
                    res = flow._run(x, inverse, log_det, prologue=prologue)
                    if res is None:  # no fused kernel after all: z0 by the formula, then the layer as usual
                        z0 = prologue[0] + prologue[1].exp().sqrt() * x
                        res = flow._run(z0, inverse, log_det)
                    x = res[0]
                elif fresh and i == 0:
                    x, _ = flow._run(x, inverse, log_det, overwrite=True)  # log_det = ld inside the kernel
                elif isinstance(flow, _TwoWayFlow) or (isinstance(flow, RNVP) and not inverse):
                    x, _ = flow._run(x, inverse, log_det)  # log_det += ld inside the kernel
                else:
                    x, ld = flow.inverse(x) if inverse else flow.forward(x)
                    log_det += ld
                seen.append(x)
            if timed:
                e_next = torch.cuda.Event(enable_timing=True)
                e_next.record()
                self.layer_events.append((e_prev, e_next, i, span))
            i += span
        return seen, log_det

    def forward(self, z: Tensor) -> tuple[list[Tensor], Tensor]:  # z -> x
        return self._pass(z, False)

    def inverse(self, x: Tensor) -> tuple[list[Tensor], Tensor]:  # x -> z
        return self._pass(x, True)


class _StdNormalLogProbFn(torch.autograd.Function):
    """log N(z; 0, I) with the autograd link: forward = the epilogue kernel (one pass over z), backward = -z g.
    (As torch ops -- pow, sum over dim 1, scale, shift and their backward -- this was 0.5 ms of a 4.5 ms training
    step at 2^20 rows x 64.)"""

    @staticmethod
    def forward(ctx, z):
        zc = z.detach().contiguous()
        lp = torch.empty(zc.shape[0], dtype=torch.float32, device=zc.device)
        _lib.check("mnf_gauss_logprob", _lib.load().mnf_gauss_logprob(
            zc.data_ptr(), None, lp.data_ptr(), None, zc.shape[0], zc.shape[1], _stream()))
        ctx.save_for_backward(zc)
        return lp

    @staticmethod
    def backward(ctx, grad_lp):
        (z,) = ctx.saved_tensors
        return z * (-grad_lp).unsqueeze(1)


class StandardNormal:
    """N(0, I_dim) base distribution whose ``log_prob`` is the HIP epilogue kernel.

    Equivalent to the ``MultivariateNormal(zeros(d), eye(d))`` the reference's tests and
    notebooks pair with the flows (tests/test_flows.py:38)."""

    def __init__(self, dim: int, device: torch.device | str = "cuda") -> None:
        self.dim = int(dim)
        self.device = torch.device(device)

    def log_prob(self, z: Tensor) -> Tensor:
        if torch.is_grad_enabled() and z.requires_grad:  # training: keep the autograd link
            if z.is_cuda and z.dim() == 2 and z.shape[0] > 0 and z.dtype == torch.float32:
                return _StdNormalLogProbFn.apply(z)
            return -0.5 * z.pow(2).sum(1) - 0.5 * self.dim * math.log(2 * math.pi)
        z = _device_input(z, "z")
        lp = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        if z.shape[0] == 0:
            return lp
        _lib.check("mnf_gauss_logprob", _lib.load().mnf_gauss_logprob(
            z.data_ptr(), None, lp.data_ptr(), None, z.shape[0], self.dim, _stream()))
        return lp

    def sample(self, sample_shape=torch.Size()) -> Tensor:
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        return torch.randn(*sample_shape, self.dim, device=self.device)


class NormalizingFlowModel(NormalizingFlow):
    """(base distribution, flows) pair (flows/core.py:38-55), plus a fused ``log_prob``."""

    def __init__(self, base, flows: Sequence[nn.Module]) -> None:
        super().__init__(flows)
        self.base = base

    def base_log_prob(self, x: Tensor) -> Tensor:
        zs, _ = self.inverse(x)
        return self.base.log_prob(zs[-1])

    def sample(self, *num_samples: int) -> Tensor:
        z = self.base.sample(*num_samples)
        xs, _ = self.forward(z)
        return xs[-1]

    def graphed_log_prob(self, example: Tensor):
        """Capture one ``log_prob(x, return_sum=True)`` pass (every coupling kernel + the epilogue)
        into a HIP graph and return ``replay(x) -> (log_prob, sum)``.

        The C ABI never allocates or synchronises, so the whole pass is capturable; replaying it
        removes the per-launch host work for small batches (at 4,096 rows the nine-layer pass is
        launch-bound).  The returned tensors are the graph's static outputs: they are overwritten
        by the next replay.  Parameters must not change between capture and replay."""
        static_x = example.detach().clone()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up: packs the parameter images outside the capture
                    self.log_prob(static_x, return_sum=True)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.log_prob(static_x, return_sum=True)

        def replay(x: Tensor):
            static_x.copy_(x)
            graph.replay()
            return out

        replay.graph = graph
        return replay

    def log_prob(self, x: Tensor, return_sum: bool = False):
        """log p(x) = log_det + base.log_prob(z) with ONE inverse pass (the reference's callers
        run two, core.py:46-49 vs examples/half_moons.ipynb:183-184).  With a StandardNormal
        base the epilogue kernel also produces the fp64 sum over rows."""
        std = isinstance(self.base, StandardNormal)
        lp = total = None
        if (std and torch.is_grad_enabled() and not _dispatch.NO_FUSED_LOGPROB and not _NO_RUN_FUSION_ENV
                and self.fuse_affine_runs and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2
                and x.shape[0] >= _dispatch.BWD_SPLIT_MIN_ROWS and x.dtype == torch.float32 and self.layer_events is None):
            # training, the whole model ONE run of AffineHalfFlow layers: one autograd node from x to log p -- the
            # stack kernel's epilogue writes log p, the gradient kernel of the last layer forms -z g itself
            runs = self._affine_runs()
            run = runs.get(0) if len(runs) == 1 else None
            if isinstance(run, _AffineRun) and len(run.layers) == len(self.flows) and run.trainable(x):
                lp_fused = run.launch_grad(x, True, with_lp=True)
                if lp_fused is not None:
                    self._last_sqnorm, self._logprob_done = None, False
                    return (lp_fused, lp_fused.detach().double().sum().reshape(1)) if return_sum else lp_fused
        if std and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0:
            lp = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
            total = torch.zeros(1, dtype=torch.float64, device=x.device) if return_sum else None
        zs, log_det = self._pass(x, True, want_sqnorm=std, want_logprob=(lp, total) if lp is not None else None,
                                 last_only=True, lp_tail=std and self.layer_events is None)
        if self._lp_node is not None:  # the closing [Glow, ActNorm] pair's node produced log p (training)
            lp_node, self._lp_node = self._lp_node, None
            return (lp_node, lp_node.detach().double().sum().reshape(1)) if return_sum else lp_node
        z = zs[-1]
        if self._logprob_done:  # the last coupling launch already produced log p and its sum
            return (lp, total) if return_sum else lp
        if std and not (torch.is_grad_enabled() and (z.requires_grad or log_det.requires_grad)):
            if lp is None:
                lp = torch.empty_like(log_det)
                total = torch.zeros(1, dtype=torch.float64, device=z.device) if return_sum else None
            if z.shape[0] == 0:
                return (lp, total) if return_sum else lp
            if self._last_sqnorm is not None:  # |z|^2 came out of the last coupling kernel
                _lib.check("mnf_gauss_logprob_sq", _lib.load().mnf_gauss_logprob_sq(
                    self._last_sqnorm.data_ptr(), log_det.data_ptr(), lp.data_ptr(), _ptr(total),
                    z.shape[0], z.shape[1], _stream()))
            else:
                _lib.check("mnf_gauss_logprob", _lib.load().mnf_gauss_logprob(
                    z.data_ptr(), log_det.data_ptr(), lp.data_ptr(), _ptr(total), z.shape[0], z.shape[1],
                    _stream()))
            return (lp, total) if return_sum else lp
        lp = log_det + self.base.log_prob(z)
        return (lp, lp.double().sum().reshape(1)) if return_sum else lp
