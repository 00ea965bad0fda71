"""The MNF caller of the path: ``MNFLinear`` (torch_mnf/layers/mnf_linear.py:7-90).

``sample_z`` is on the hot path: it draws the multiplicative noise
``z0 = q0_mean + sigma * eps`` for every row of the batch and pushes it through ``flow_q``
(a stack of masked/gated ``RNVP`` layers) -- at MNF-LeNet's 512 images x 500 MC samples that is
256,000 rows of 800 dims.  Both steps run in libmnf_hip.so, and so does what ``forward`` does with the
result: the two products of the local reparametrisation and the noise epilogue are one
launch that reads x and z once (``mnf_mnf_linear_fwd``).  ``kl_div`` runs both flows and then every closed-form
term of the reference's formula in one launch each way (``mnf_mnf_kl_fwd`` / ``_bwd``).
"""
from __future__ import annotations

import ctypes
import math
import os

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import _lib
from . import flows as _flows
from ._lib import MnfHipError
from .flows import MADE, RNVP, MaskedLinear, NormalizingFlow, _stream, _wants_grad  # noqa: F401 (MADE, MaskedLinear: torch_mnf.layers exports them)


def _mnf_linear_forward(module, x, z, eps, seed, ops, sd):
    """One mnf_mnf_linear_fwd(_train) launch: (out, range-flag workspace)."""
    flat, image, var_unscale = ops
    rows = x.shape[0]
    out = torch.empty(rows, module.n_out, dtype=torch.float32, device=x.device)
    work = torch.empty((rows + 127) // 128, dtype=torch.int32, device=x.device)
    _lib.check("mnf_mnf_linear_fwd_train", _lib.load().mnf_mnf_linear_fwd_train(
        x.data_ptr(), z.data_ptr(), None if eps is None else eps.data_ptr(), seed, out.data_ptr(),
        None if sd is None else sd.data_ptr(), flat.data_ptr(), image.data_ptr(), var_unscale, work.data_ptr(), rows,
        module.n_in, module.n_out, _stream()))
    return out, work


def _require_operands(module, device):
    """The forward launch's packed operands, or a clear error when the library has no kernel for the shape (the split
    layout refuses n_in * 512 >= 2^30): every caller unpacks the triple."""
    ops = module._forward_operands(device)
    if ops is None:
        raise _lib.MnfHipError("mnf_mnf_linear_split_layout", _lib.MNF_ERR_UNSUPPORTED,
                               f"MNFLinear({module.n_in}, {module.n_out}): no HIP kernel for this shape")
    return ops


_MNF_LINEAR_BWD_WORK: dict = {}  # (device, stream) -> scratch of mnf_mnf_linear_bwd (its launches follow one another on a stream)


class _MnfLinearFn(torch.autograd.Function):
    """MNFLinear.forward with gradients: the forward launch keeps sqrt(var); backward = mnf_mnf_linear_bwd."""

    @staticmethod
    def forward(ctx, x, z, W_mean, W_log_var, b_mean, b_log_var, module, eps, seed):
        ops = _require_operands(module, x.device)
        xc, zc = x.detach().contiguous(), z.detach().contiguous()
        sd = torch.empty(xc.shape[0], module.n_out, dtype=torch.float32, device=x.device)
        out, work = _mnf_linear_forward(module, xc, zc, eps, seed, ops, sd)
        ctx.module, ctx.seed, ctx.var_unscale = module, seed, ops[2]
        # the four parameters in one train.FlatParameters buffer: backward adds to their gradient slice in place
        ctx.home = _flows._flat_home_of(module, [W_mean, W_log_var, b_mean, b_log_var])
        ctx.save_for_backward(xc, zc, sd, work, ops[0], *([eps] if eps is not None else []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.module
        xc, zc, sd, fwd_flags, flat, *rest = ctx.saved_tensors
        eps = rest[0] if rest else None
        lib = _lib.load()
        rows, dev = xc.shape[0], xc.device
        g = grad_out.contiguous()
        idx, n_split = m._bwd_index(dev)
        image = torch.empty(n_split + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32, device=dev)
        _lib.check("mnf_pack_gather_split", lib.mnf_pack_gather_split(
            flat.data_ptr(), idx.data_ptr(), image.data_ptr(), n_split, 0, _stream()))
        need = int(lib.mnf_mnf_linear_bwd_workspace_bytes(rows, m.n_in, m.n_out))
        key = (dev, _stream())
        work = _MNF_LINEAR_BWD_WORK.get(key)
        if work is None or work.numel() < need:
            work = torch.empty(need, dtype=torch.uint8, device=dev)
            _MNF_LINEAR_BWD_WORK[key] = work
        scale = _flows._grad_scale(g, None, rows, m.n_out, dev)
        grad_x, grad_z = torch.empty_like(xc), torch.empty_like(zc)
        home = ctx.home
        grad_flat = home[0].grad[home[1]:home[1] + home[2]] if home is not None else torch.zeros_like(flat)
        _lib.check("mnf_mnf_linear_bwd", lib.mnf_mnf_linear_bwd(
            xc.data_ptr(), zc.data_ptr(), g.data_ptr(), sd.data_ptr(), None if eps is None else eps.data_ptr(), ctx.seed,
            grad_x.data_ptr(), grad_z.data_ptr(), grad_flat.data_ptr(), flat.data_ptr(), image.data_ptr(), ctx.var_unscale,
            fwd_flags.data_ptr(), scale.data_ptr(), work.data_ptr(), work.numel(), rows, m.n_in, m.n_out, _stream()))
        if home is not None:
            return grad_x, grad_z, None, None, None, None, None, None, None
        n = m.n_in * m.n_out
        gW_mean = grad_flat[:n].view(m.n_out, m.n_in)
        gW_log_var = grad_flat[n:2 * n].view(m.n_out, m.n_in)
        gb_mean = grad_flat[2 * n:2 * n + m.n_out]
        gb_log_var = grad_flat[2 * n + m.n_out:2 * n + 2 * m.n_out]
        return grad_x, grad_z, gW_mean, gW_log_var, gb_mean, gb_log_var, None, None, None


def _flow_through(flow: NormalizingFlow, z: Tensor, masks):
    """``flow.forward`` on a (1, n_out) row, with explicit masks when given: (last z, log_det (1,))."""
    if masks is None:
        zs, log_det = flow.forward(z)
        return zs[-1], log_det
    layers = list(flow.flows)
    if len(masks) != len(layers):
        raise ValueError(f"got {len(masks)} masks for {len(layers)} flow layers")
    log_det = torch.zeros(z.shape[0], device=z.device)
    for layer, m in zip(layers, masks):
        if _wants_grad(layer, z):
            z, ld = layer._run(z, False, None, m)
            log_det = log_det + ld
        else:
            z, _ = layer._run(z, False, log_det, m)
    return z, log_det


class _SampleZ0Fn(torch.autograd.Function):
    """z0 = q0_mean + sqrt(exp(q0_log_var)) * eps (mnf_linear.py:59-62, mnf_conv.py:91-93) with its two parameter
    gradients from the library: one launch each way instead of four elementwise kernels forward and ~eight backward.
    ``eps`` None: the noise is the library's counter-based N(0, 1) of (``seed``, row, column), generated inside both
    launches (``mnf_sample_z0_noise(seed, ., rows, n_in)`` materialises it): no (rows, n_in) noise tensor exists."""

    @staticmethod
    def forward(ctx, q0_mean, q0_log_var, eps, module, seed=0, rows=0):
        mean, log_var = q0_mean.detach().contiguous(), q0_log_var.detach().contiguous()
        if eps is None:
            z0 = torch.empty(rows, mean.numel(), dtype=torch.float32, device=mean.device)
            _lib.check("mnf_sample_z0_seeded", _lib.load().mnf_sample_z0_seeded(
                mean.data_ptr(), log_var.data_ptr(), seed, z0.data_ptr(), rows, mean.numel(), _stream()))
            ctx.save_for_backward(log_var)
        else:
            z0 = torch.empty_like(eps)
            _lib.check("mnf_sample_z0", _lib.load().mnf_sample_z0(
                mean.data_ptr(), log_var.data_ptr(), eps.data_ptr(), z0.data_ptr(), eps.shape[0], eps.shape[1], _stream()))
            ctx.save_for_backward(log_var, eps)
        ctx.seed = seed
        ctx.home = _flows._flat_home_of(module, [q0_mean, q0_log_var])
        return z0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_z0):
        log_var, *rest = ctx.saved_tensors
        eps = rest[0] if rest else None
        dim, home = log_var.numel(), ctx.home
        out = home[0].grad[home[1]:home[1] + home[2]] if home is not None else torch.zeros(2 * dim, device=log_var.device)
        g = grad_z0.contiguous()
        if _lib.deterministic():  # MNF_DETERMINISTIC=1: the sums over the rows in a fixed order
            n_ws = _lib.load().mnf_sample_z0_bwd_workspace(g.shape[0], dim)
            ws = torch.empty(max(int(n_ws), 1), dtype=torch.float32, device=g.device)
            _lib.check("mnf_sample_z0_bwd_det", _lib.load().mnf_sample_z0_bwd_det(
                g.data_ptr(), None if eps is None else eps.data_ptr(), ctx.seed, log_var.data_ptr(), out.data_ptr(),
                out.data_ptr() + 4 * dim, g.shape[0], dim, ws.data_ptr(), ws.numel(), _stream()))
        elif eps is None:
            _lib.check("mnf_sample_z0_seeded_bwd", _lib.load().mnf_sample_z0_seeded_bwd(
                g.data_ptr(), ctx.seed, log_var.data_ptr(), out.data_ptr(), out.data_ptr() + 4 * dim, g.shape[0], dim,
                _stream()))
        else:
            _lib.check("mnf_sample_z0_bwd", _lib.load().mnf_sample_z0_bwd(
                g.data_ptr(), eps.data_ptr(), log_var.data_ptr(), out.data_ptr(), out.data_ptr() + 4 * dim, eps.shape[0],
                dim, _stream()))
        if home is not None:
            return None, None, None, None, None, None
        return out[:dim], out[dim:], None, None, None, None


class _ConvOperandsFn(torch.autograd.Function):
    """(W_mean * z per output channel, exp(W_log_var), exp(b_log_var)) -- the weights and bias of MNFConv2d.forward's two
    convolutions (mnf_conv.py:69-72) -- one launch each way; the three parameters' gradients are added in place when
    they live in a train.FlatParameters buffer."""

    @staticmethod
    def forward(ctx, W_mean, W_log_var, b_log_var, z, module):
        Wm, Wl, bl, zc = (t.detach().contiguous() for t in (W_mean, W_log_var, b_log_var, z.reshape(-1)))
        n_out, per_out = Wm.shape[0], Wm[0].numel()
        Wz, Wv, bv = torch.empty_like(Wm), torch.empty_like(Wm), torch.empty_like(bl)
        _lib.check("mnf_mnf_conv_operands", _lib.load().mnf_mnf_conv_operands(
            Wm.data_ptr(), Wl.data_ptr(), bl.data_ptr(), zc.data_ptr(), Wz.data_ptr(), Wv.data_ptr(), bv.data_ptr(),
            n_out, per_out, _stream()))
        ctx.save_for_backward(Wm, Wl, bl, zc)
        ctx.z_shape = tuple(z.shape)
        params = [W_mean, W_log_var, b_log_var]
        ctx.home = _flows._flat_home_of(module, params) if all(p.requires_grad for p in params) else None
        ctx.set_materialize_grads(False)
        return Wz, Wv, bv

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_wz, g_wv, g_bv):
        Wm, Wl, bl, zc = ctx.saved_tensors
        n_out, per_out, home = Wm.shape[0], Wm[0].numel(), ctx.home
        n = n_out * per_out
        buf = home[0].grad[home[1]:home[1] + home[2]] if home is not None else torch.empty(2 * n + n_out, device=Wm.device)
        gz = torch.empty(n_out, device=Wm.device)
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        g_wz, g_wv, g_bv = cont(g_wz), cont(g_wv), cont(g_bv)
        _lib.check("mnf_mnf_conv_operands_bwd", _lib.load().mnf_mnf_conv_operands_bwd(
            Wm.data_ptr(), Wl.data_ptr(), bl.data_ptr(), zc.data_ptr(), _flows._ptr(g_wz), _flows._ptr(g_wv),
            _flows._ptr(g_bv), buf.data_ptr(), buf.data_ptr() + 4 * n, buf.data_ptr() + 8 * n, gz.data_ptr(), n_out,
            per_out, int(home is not None), _stream()))
        gz = gz.view(ctx.z_shape)
        if home is not None:
            return None, None, None, gz, None
        return buf[:n].view_as(Wm), buf[n:2 * n].view_as(Wm), buf[2 * n:], gz, None


class _NoiseFn(torch.autograd.Function):
    """mean + sqrt(var) * eps (mnf_conv.py:86-88; the local-reparametrisation draw), one launch each way."""

    @staticmethod
    def forward(ctx, mean, var, eps):
        m, v, e = mean.detach().contiguous(), var.detach().contiguous(), eps.detach().contiguous()
        out = torch.empty_like(m)
        _lib.check("mnf_mnf_noise", _lib.load().mnf_mnf_noise(m.data_ptr(), v.data_ptr(), e.data_ptr(), out.data_ptr(),
                                                             m.numel(), _stream()))
        ctx.save_for_backward(v, e)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        v, e = ctx.saved_tensors
        gc = g.contiguous()
        gv = torch.empty_like(v)
        _lib.check("mnf_mnf_noise_bwd", _lib.load().mnf_mnf_noise_bwd(v.data_ptr(), e.data_ptr(), gc.data_ptr(),
                                                                     gv.data_ptr(), v.numel(), _stream()))
        return gc, gv, None


class _MnfKlFn(torch.autograd.Function):
    """``kl_div`` of either MNF layer behind its flows (mnf_linear.py:66-90, mnf_conv.py:100-133): one launch forward
    (``mnf_mnf_kl_fwd``) and one backward (``mnf_mnf_kl_bwd``) instead of ~70 + ~100 elementwise kernels.  The random
    draws (``eps``, ``eps_b``) and both flows' outputs are inputs; see include/mnf_hip.h for the (rows, cols) view.
    ``params``: the layer's own parameters in registration order (W_mean, W_log_var, [b_mean,] b_log_var, q0_mean,
    q0_log_var, r0_c, r0_b1, r0_b2) -- also the layout of the kernel's parameter gradients, so that a layer living in a
    train.FlatParameters buffer has them added to its gradient slice in place."""

    @staticmethod
    def forward(ctx, z, log_det_q, z_r, log_det_r, eps, eps_b, module, conv, b_mean, *params):
        W_mean, W_log_var = params[0], params[1]
        if conv:
            b_log_var, _, q0_log_var, r0_c, r0_b1, r0_b2 = params[2:]
        else:
            b_mean, b_log_var, _, q0_log_var, r0_c, r0_b1, r0_b2 = params[2:]
        dev = W_mean.device
        if dev.type != "cuda":
            raise MnfHipError("mnf_mnf_kl_fwd", _lib.MNF_ERR_NO_DEVICE,
                              "kl_div runs in libmnf_hip.so: the layer must live on a HIP device")
        cols = r0_c.numel()
        rows = W_mean.numel() // cols
        n_bias = b_log_var.numel()

        def f32(t, n, name):
            if t is None:
                return None
            if t.device != dev or t.numel() != n:
                raise ValueError(f"kl_div: {name} must hold {n} elements on {dev}, got {tuple(t.shape)} on {t.device}")
            return t.detach().to(torch.float32).contiguous()

        ops = [f32(W_mean, rows * cols, "W_mean"), f32(W_log_var, rows * cols, "W_log_var"),
               f32(eps, rows if conv else rows * cols, "eps"), f32(eps_b, 1, "eps_b") if conv else None,
               f32(z, cols, "z"), f32(z_r, cols, "flow_r's output"), f32(log_det_q, 1, "log_det_q"),
               f32(log_det_r, 1, "log_det_r"), f32(b_mean, n_bias, "b_mean"), f32(b_log_var, n_bias, "b_log_var"),
               f32(q0_log_var, cols, "q0_log_var"), f32(r0_c, cols, "r0_c"), f32(r0_b1, cols, "r0_b1"),
               f32(r0_b2, cols, "r0_b2")]
        lib = _lib.load()
        out = torch.empty((), device=dev)
        saved = torch.empty(lib.mnf_mnf_kl_saved_floats(rows), device=dev)
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        _lib.check("mnf_mnf_kl_fwd", lib.mnf_mnf_kl_fwd(*[ptr(t) for t in ops], int(conv), rows, cols, n_bias,
                                                       out.data_ptr(), saved.data_ptr(), _stream()))
        ctx.ops, ctx.saved_acts, ctx.shape = ops, saved, (int(conv), rows, cols, n_bias)
        ctx.like = tuple(tuple(t.shape) for t in (z, log_det_q, z_r, log_det_r))  # (shapes only: no references kept)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.home = _flows._flat_home_of(module, list(params)) if all(p.requires_grad for p in params) else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        conv, rows, cols, n_bias = ctx.shape
        W_mean, W_log_var, eps, eps_b, z, z_r, _, _, b_mean, b_log_var, _, c, b1, b2 = ctx.ops
        z_in, ldq_in, zr_in, ldr_in = ctx.like
        lib = _lib.load()
        dev = W_mean.device
        grads = torch.empty(lib.mnf_mnf_kl_grad_floats(cols), device=dev)
        n_param = lib.mnf_mnf_kl_param_grad_floats(conv, rows, cols, n_bias)
        home = ctx.home
        if home is not None and home[2] != n_param:
            raise MnfHipError("mnf_mnf_kl_bwd", _lib.MNF_ERR_INVALID_ARG,
                              f"the layer's parameter slice holds {home[2]} floats, the kernel writes {n_param}")
        pg = home[0].grad[home[1]:home[1] + home[2]] if home is not None else torch.empty(n_param, device=dev)
        g = grad_out.detach().to(torch.float32).contiguous()
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        _lib.check("mnf_mnf_kl_bwd", lib.mnf_mnf_kl_bwd(
            *[ptr(t) for t in (W_mean, W_log_var, eps, eps_b, z, z_r, b_mean, b_log_var, c, b1, b2)],
            ctx.saved_acts.data_ptr(), g.data_ptr(), conv, rows, cols, n_bias, grads.data_ptr(), pg.data_ptr(),
            int(home is not None), _stream()))
        gz, gzr, gldq, gldr = torch.split(grads, [cols, cols, 1, 1])
        head = (gz.view(z_in), gldq.view(ldq_in), gzr.view(zr_in), gldr.view(ldr_in), None, None, None, None, None)
        if home is not None:
            return head + (None,) * len(ctx.param_shapes)
        sizes = [math.prod(sh) for sh in ctx.param_shapes]
        return head + tuple(t.view(sh) for t, sh in zip(torch.split(pg, sizes), ctx.param_shapes))


class _OutputSlab:
    """Outputs [lo, hi) of an MNFLinear wider than 64 outputs, seen as a layer of its own by the <= 64-output kernels
    (mnf_mnf_linear_fwd / _bwd): the parameter attributes are fresh slices of the parent's parameters (autograd views:
    the slab's gradients flow back into the parent's through the slice), the operand caches live on this object."""

    def __init__(self, parent: "MNFLinear", lo: int, hi: int) -> None:
        self.parent, self.lo, self.hi = parent, lo, hi
        self.n_in, self.n_out = parent.n_in, hi - lo

    W_mean = property(lambda self: self.parent.W_mean[self.lo:self.hi])
    W_log_var = property(lambda self: self.parent.W_log_var[self.lo:self.hi])
    b_mean = property(lambda self: self.parent.b_mean[self.lo:self.hi])
    b_log_var = property(lambda self: self.parent.b_log_var[self.lo:self.hi])


class MNFLinear(nn.Module):
    """Bayesian linear layer with multiplicative normalizing-flow noise.

    Same constructor, parameter names and state_dict keys as the reference."""

    def __init__(self, n_in: int, n_out: int, n_flows_q: int = 2, n_flows_r: int = 2, h_sizes=(50,)) -> None:
        super().__init__()
        self.n_in, self.n_out = int(n_in), int(n_out)
        small = lambda *shape: 0.1 * torch.randn(*shape)            # noqa: E731
        log_var = lambda *shape: -9 + 0.1 * torch.randn(*shape)     # noqa: E731
        self.W_mean = nn.Parameter(small(n_out, n_in))
        self.W_log_var = nn.Parameter(log_var(n_out, n_in))
        self.b_mean = nn.Parameter(torch.zeros(n_out))
        self.b_log_var = nn.Parameter(log_var(n_out))
        self.q0_mean = nn.Parameter(small(n_in))
        self.q0_log_var = nn.Parameter(log_var(n_in))
        self.r0_c = nn.Parameter(small(n_in))
        self.r0_b1 = nn.Parameter(small(n_in))
        self.r0_b2 = nn.Parameter(small(n_in))
        self.flow_q = NormalizingFlow([RNVP(n_in, h_sizes=h_sizes) for _ in range(n_flows_q)])
        self.flow_r = NormalizingFlow([RNVP(n_in, h_sizes=h_sizes) for _ in range(n_flows_r)])
        self.fuse_prologue = True  # sample_z: form z0 inside the first flow's kernel when that kernel exists

    def _fused_prologue_ok(self, flow, eps) -> bool:
        """The first flow's split MFMA kernel can form z0 in its loads (49 <= d <= 1024, h <= 64).  Inference
        only: when any flow_q layer wants gradients the pass goes through the layers' autograd functions."""
        if self.fuse_prologue is False or (torch.is_grad_enabled()
                                           and any(p.requires_grad for p in self.flow_q.parameters())):
            return False
        return (not flow.force_generic and self.n_in <= 1024 and flow._packed(eps.device)[1] is not None
                and flow._split_image(eps.device) is not None)

    # ------------------------------------------------------------------ hot path
    def sample_z(self, batch_size: int = 1, eps: Tensor | None = None, masks=None) -> tuple[Tensor, Tensor]:
        """(mnf_linear.py:58-64).  ``eps`` (batch, n_in) and ``masks`` (one per flow_q layer) may be
        injected for reproducible runs; by default both are drawn on the device."""
        dev = self.q0_mean.device
        training = torch.is_grad_enabled() and (self.q0_mean.requires_grad or self.q0_log_var.requires_grad)
        flows = list(self.flow_q.flows)
        if masks is not None and len(masks) != len(flows):
            raise ValueError(f"sample_z got {len(masks)} masks for {len(flows)} flow_q layers")
        if (eps is None and training and batch_size > 0 and dev.type == "cuda" and not _flows._DEVICE_MASKS
                and not torch.cuda.is_current_stream_capturing()):
            # training, nothing injected: the noise is generated inside the prologue launch and again inside its gradient
            # launch from one host-drawn seed (never as a (batch, n_in) tensor).  (A step being recorded in a hipGraph
            # -- train.GraphedStep's, or a user's own torch.cuda.graph capture -- keeps the tensor draw: a seed would be
            # frozen into the recorded kernel arguments and every replay would use the same noise.)
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
            z0 = _SampleZ0Fn.apply(self.q0_mean, self.q0_log_var, None, self, seed, int(batch_size))
            return self._through_flow_q(z0, masks, dev)
        if eps is None:
            eps = torch.randn(batch_size, self.n_in, device=dev)
        eps = eps.to(dev, torch.float32).contiguous()
        if (not training and eps.shape[0] > 0 and flows and isinstance(flows[0], RNVP)
                and self._fused_prologue_ok(flows[0], eps)):
            # the prologue z0 = q0_mean + q0_std eps is formed inside the first flow's kernel: z0 is never stored
            prologue = (self.q0_mean.detach().contiguous(), self.q0_log_var.detach().contiguous())
            if masks is None:
                zs, log_det = self.flow_q._pass(eps, False, prologue=prologue)
                return zs[-1], log_det.squeeze()
            log_det = torch.zeros(eps.shape[0], device=dev)
            res = flows[0]._run(eps, False, log_det, masks[0], prologue=prologue)
            if res is not None:
                z = res[0]
                for flow, m in zip(flows[1:], masks[1:]):
                    z, _ = flow._run(z, False, log_det, m)
                return z, log_det.squeeze()
        if training:
            z0 = _SampleZ0Fn.apply(self.q0_mean, self.q0_log_var, eps, self)
        else:
            z0 = torch.empty_like(eps)
            if eps.shape[0] > 0:
                _lib.check("mnf_sample_z0", _lib.load().mnf_sample_z0(
                    self.q0_mean.detach().contiguous().data_ptr(), self.q0_log_var.detach().contiguous().data_ptr(),
                    eps.data_ptr(), z0.data_ptr(), eps.shape[0], self.n_in, _stream()))
        return self._through_flow_q(z0, masks, dev)

    def _through_flow_q(self, z0: Tensor, masks, dev) -> tuple[Tensor, Tensor]:
        if masks is None:
            zs, log_det = self.flow_q.forward(z0)
        else:  # same loop as NormalizingFlow.forward with the masks handed to each RNVP
            log_det = torch.zeros(z0.shape[0], device=dev)
            zs = [z0]
            for flow, m in zip(self.flow_q.flows, masks):
                if _wants_grad(flow, zs[-1]):  # autograd path: flow_q gradients from the *_bwd kernels
                    z, ld = flow._run(zs[-1], False, None, m)
                    log_det = log_det + ld
                else:
                    z, _ = flow._run(zs[-1], False, log_det, m)
                zs.append(z)
        return zs[-1], log_det.squeeze()

    # ------------------------------------------------------------------ forward behind the flow path
    def _forward_operands(self, device):
        """(flat parameters, split operand image, var_unscale) of mnf_mnf_linear_fwd for the current parameters, or
        None when the shape has no kernel (n_out > 64).  Repacked when a parameter changes."""
        params = (self.W_mean, self.W_log_var, self.b_mean, self.b_log_var)
        # (an _OutputSlab borrows this method: its parameters are slices of its PARENT's, and it is the parent that a
        #  train.FlatParameters buffer re-homes -- a fused optimizer step or a hipGraph replay writes that buffer through
        #  raw pointers and only bumps the home's generation, never a parameter's version counter)
        flat_home = getattr(self, "parent", self).__dict__.get("_mnf_flat")
        key = (device, 0 if flat_home is None else flat_home.generation,
               tuple((p.data_ptr(), p._version) for p in params))
        cache = self.__dict__.get("_fwd_cache")
        if cache is None or cache[0] != key:
            lib = _lib.load()
            n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
            rc = lib.mnf_mnf_linear_split_layout(self.n_in, self.n_out, ctypes.byref(n_split), ctypes.byref(n_plain))
            if rc == _lib.MNF_ERR_UNSUPPORTED:
                cache = (key, None)
            else:
                _lib.check("mnf_mnf_linear_split_layout", rc)
                index = self.__dict__.get("_fwd_index")
                if index is None or index.device != device:
                    idx = (ctypes.c_int32 * (2 * n_split.value + n_plain.value))()
                    _lib.check("mnf_mnf_linear_split_index", lib.mnf_mnf_linear_split_index(self.n_in, self.n_out, idx))
                    index = torch.frombuffer(idx, dtype=torch.int32).clone().to(device)
                    self.__dict__["_fwd_index"] = index
                with torch.no_grad():
                    w_var = self.W_log_var.detach().to(device, torch.float32).exp()
                    # exp(W_log_var) is ~1e-4 at init and shrinks in training: times a power of two that puts its
                    # largest entry near [0.5, 1), so that the f16 halves of the image are normal numbers.  The power
                    # is read back from the device (one host synchronisation) at the first pack and re-read every 64th
                    # pack after that -- never while a hipGraph is being captured: log-variances drift by a few powers
                    # of two over a training run, f16 has 2^-14 .. 2^15 of normal range around the target, and a
                    # weight that does leave the range sends the launch to the fp32 path (the image's weight guard).
                    shift = self.__dict__.get("_var_shift")
                    packs = self.__dict__.get("_var_shift_age", 0)
                    if shift is None or (packs >= 64 and not torch.cuda.is_current_stream_capturing()):
                        top = float(w_var.max())
                        shift = -math.frexp(top)[1] if top > 0.0 and math.isfinite(top) else 0
                        shift = max(min(shift, 100), -100)
                        self.__dict__["_var_shift"], packs = shift, 0
                    self.__dict__["_var_shift_age"] = packs + 1
                    flat = torch.cat([self.W_mean.detach().to(device, torch.float32).reshape(-1),
                                      (w_var * (2.0 ** shift)).reshape(-1),
                                      self.b_mean.detach().to(device, torch.float32),
                                      self.b_log_var.detach().to(device, torch.float32).exp()]).contiguous()
                image = torch.empty(n_split.value + n_plain.value + _lib.MNF_SPLIT_TAIL_WORDS, dtype=torch.int32,
                                    device=device)
                _lib.check("mnf_pack_gather_split", lib.mnf_pack_gather_split(
                    flat.data_ptr(), index.data_ptr(), image.data_ptr(), n_split.value, n_plain.value, _stream()))
                cache = (key, (flat, image, 2.0 ** -shift))
                if _flows._CHECK_PARAMS_EVERY:  # debug switch: the raw parameters the image was packed from
                    self.__dict__["_fwd_packed_from"] = torch.cat([p.detach().reshape(-1) for p in params]).clone()
            self.__dict__["_fwd_cache"] = cache
        elif _flows._CHECK_PARAMS_EVERY:
            _flows._check_params_fresh(params, self.__dict__.get("_fwd_packed_from"), "MNFLinear.forward")
        return cache[1]

    def _bwd_index(self, device):
        """(device index table, n_split_words) of the gradient kernels' operand image."""
        cached = self.__dict__.get("_bwd_idx")
        if cached is None or cached[0].device != device:
            lib = _lib.load()
            n_split, n_plain = ctypes.c_int64(0), ctypes.c_int64(0)
            _lib.check("mnf_mnf_linear_bwd_layout", lib.mnf_mnf_linear_bwd_layout(
                self.n_in, self.n_out, ctypes.byref(n_split), ctypes.byref(n_plain)))
            idx = (ctypes.c_int32 * (2 * n_split.value))()
            _lib.check("mnf_mnf_linear_bwd_index", lib.mnf_mnf_linear_bwd_index(self.n_in, self.n_out, idx))
            cached = (torch.frombuffer(idx, dtype=torch.int32).clone().to(device), n_split.value)
            self.__dict__["_bwd_idx"] = cached
        return cached

    def forward(self, x: Tensor, eps: Tensor | None = None) -> Tensor:
        """Algorithm 1 of the MNF paper (mnf_linear.py:46-56): ``mean + sqrt(var) * eps`` with
        ``mean = (x * z) @ W_mean.T + b_mean`` and ``var = x**2 @ exp(W_log_var).T + exp(b_log_var)``.

        Both products and the noise epilogue are ONE HIP launch behind ``sample_z`` (x and z read once; ``eps``
        (rows, n_out) may be injected, by default it is generated inside the kernel from a seed drawn from torch's
        generator).  With gradients wanted the same launch also keeps ``sqrt(var)``, and the backward pass is
        ``mnf_mnf_linear_bwd`` (grad x, grad z, dW_mean, dW_log_var, db_mean, db_log_var on the matrix cores): no
        stock-PyTorch matrix product anywhere on the path.  A layer wider than 64 outputs runs the same launches once
        per 64-output slab."""
        if x.dim() != 2 or x.shape[1] != self.n_in:
            raise ValueError(f"MNFLinear({self.n_in}, {self.n_out}) expects (rows, {self.n_in}) inputs, got {tuple(x.shape)}")
        z, _ = self.sample_z(x.size(0))
        if not x.is_cuda or z.device != x.device:
            raise RuntimeError(f"torch_mnf_amd: MNFLinear.forward needs the input ({x.device}) and the layer "
                               f"({z.device}) on the same GPU; the HIP path has no CPU fallback")
        if x.dtype != torch.float32:
            raise TypeError(f"torch_mnf_amd: MNFLinear.forward needs float32 inputs, got {x.dtype}")
        if x.shape[0] == 0:
            return x.new_empty(0, self.n_out)
        seed = 0
        if eps is None and _flows._DEVICE_MASKS:
            # a step being recorded in a hipGraph (train.GraphedStep): a host-drawn seed would be frozen into the
            # kernel arguments and every replay would add the same noise; torch.randn is redrawn by every replay
            eps = torch.randn(x.shape[0], self.n_out, device=x.device)
        elif eps is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
        else:
            eps = eps.detach().to(x.device, torch.float32).contiguous()
            if eps.shape != (x.shape[0], self.n_out):
                raise ValueError(f"eps must be {(x.shape[0], self.n_out)}, got {tuple(eps.shape)}")
        params = (self.W_mean, self.W_log_var, self.b_mean, self.b_log_var)
        training = torch.is_grad_enabled() and (x.requires_grad or z.requires_grad or any(p.requires_grad for p in params))
        if self.n_out > 64:
            # the kernels hold one row's outputs in <= 4 accumulator tiles: a wider layer (mnf_linear.py:46-56 has no
            # width limit; an ordinary MNFFeedForward([784, 256, 10])) runs them once per 64-output slab -- x and z are
            # re-read per slab, the slabs' gradients for x and z are summed by autograd
            outs = []
            xc, zc = (None, None) if training else (x.detach().contiguous(), z.detach().contiguous())
            for k, slab in enumerate(self._output_slabs()):
                eps_k = None if eps is None else eps[:, slab.lo:slab.hi].contiguous()
                seed_k = self._slab_seed(seed, k)
                if training:
                    outs.append(_MnfLinearFn.apply(x, z, slab.W_mean, slab.W_log_var, slab.b_mean, slab.b_log_var, slab,
                                                   eps_k, seed_k))
                else:
                    outs.append(_mnf_linear_forward(slab, xc, zc, eps_k, seed_k, _require_operands(slab, x.device), None)[0])
            return torch.cat(outs, dim=1)
        if training:
            return _MnfLinearFn.apply(x, z, *params, self, eps, seed)
        ops = _require_operands(self, x.device)
        return _mnf_linear_forward(self, x.detach().contiguous(), z.detach().contiguous(), eps, seed, ops, None)[0]

    def _output_slabs(self) -> list:
        slabs = self.__dict__.get("_slabs")
        if slabs is None:
            slabs = self.__dict__["_slabs"] = [_OutputSlab(self, lo, min(lo + 64, self.n_out))
                                               for lo in range(0, self.n_out, 64)]
        return slabs

    @staticmethod
    def _slab_seed(seed: int, k: int) -> int:
        """The in-kernel noise seed of output slab k (the stream is indexed by the column INSIDE the slab)."""
        return (int(seed) + k * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    def invalidate(self) -> None:
        """Drop the packed operands of ``forward`` (they are keyed on the parameters' version counters, which a write
        through ``p.data`` or a hipGraph replay does not bump); the flows have their own ``invalidate``."""
        self.__dict__.pop("_fwd_cache", None)
        for slab in self.__dict__.get("_slabs") or ():
            slab.__dict__.pop("_fwd_cache", None)

    def noise_for(self, seed: int, rows: int, device="cuda") -> Tensor:
        """The (rows, n_out) noise an in-kernel-noise ``forward`` call with this seed used (tests)."""
        if self.n_out > 64:  # one stream per 64-output slab (forward)
            parts = []
            for k, slab in enumerate(self._output_slabs()):
                e = torch.empty(rows, slab.n_out, dtype=torch.float32, device=device)
                if rows:
                    _lib.check("mnf_mnf_linear_noise", _lib.load().mnf_mnf_linear_noise(
                        self._slab_seed(seed, k), e.data_ptr(), rows, slab.n_out, _stream()))
                parts.append(e)
            return torch.cat(parts, dim=1)
        e = torch.empty(rows, self.n_out, dtype=torch.float32, device=device)
        if rows:
            _lib.check("mnf_mnf_linear_noise", _lib.load().mnf_mnf_linear_noise(
                int(seed) & 0xFFFFFFFFFFFFFFFF, e.data_ptr(), rows, self.n_out, _stream()))
        return e

    # ------------------------------------------------------------------ the KL term behind both flows
    def kl_div(self, noise: dict | None = None) -> Tensor:
        """(mnf_linear.py:66-90): ``sample_z()`` through flow_q, the same z through flow_r, and every closed-form term
        in one launch (``mnf_mnf_kl_fwd``; gradients: ``mnf_mnf_kl_bwd``).  ``noise``: optional injected draws
        {"eps_z" (1, n_in), "masks_q", "eps_w" (n_out, n_in), "masks_r"}; by default all are drawn on the device."""
        noise = noise or {}
        z, log_det_q = self.sample_z(1, noise.get("eps_z"), noise.get("masks_q"))
        z_r, log_det_r = _flow_through(self.flow_r, z, noise.get("masks_r"))
        eps_w = noise.get("eps_w")
        if eps_w is None:
            eps_w = torch.randn_like(self.W_log_var)
        if log_det_r.numel() != 1:  # the reference unpacks `[log_det_r]` (:84)
            raise ValueError(f"kl_div: flow_r returned {log_det_r.numel()} log-determinants for one row")
        return _MnfKlFn.apply(z, log_det_q, z_r, log_det_r, eps_w, None, self, False, None, self.W_mean, self.W_log_var,
                              self.b_mean, self.b_log_var, self.q0_mean, self.q0_log_var, self.r0_c, self.r0_b1,
                              self.r0_b2)


_OutputSlab._forward_operands = MNFLinear._forward_operands
_OutputSlab._bwd_index = MNFLinear._bwd_index


class MNFConv2d(nn.Module):
    """Bayesian 2-D convolution with multiplicative normalizing-flow noise on the output channels
    (layers/mnf_conv.py:10-133): same constructor, parameter names and state_dict keys as the reference.

    What is on the library's path here is the flow: ``sample_z`` pushes one n_out-vector through ``flow_q`` (RNVP
    layers, HIP kernels) and ``kl_div`` one through ``flow_r``.  The two convolutions are the caller's arithmetic
     and stay ``F.conv2d`` on device tensors.  Noise can be injected for reproducible runs:
    ``eps_z`` (n_out,), ``masks`` (one (1, n_out) float mask per flow layer), ``eps`` (shape of the output)."""

    def __init__(self, n_in: int, n_out: int, kernel_size: int, n_flows_q: int = 2, n_flows_r: int = 2,
                 h_sizes=(50,)) -> None:
        super().__init__()
        self.n_in, self.n_out, self.kernel_size = int(n_in), int(n_out), int(kernel_size)
        small = lambda *shape: 0.1 * torch.randn(*shape)            # noqa: E731
        log_var = lambda *shape: -9 + 0.1 * torch.randn(*shape)     # noqa: E731
        w_shape = (n_out, n_in, kernel_size, kernel_size)
        self.W_mean = nn.Parameter(small(*w_shape))
        self.W_log_var = nn.Parameter(log_var(*w_shape))
        self.b_mean = torch.zeros(n_out)  # (a plain tensor in the reference too: not in the state_dict, mnf_conv.py:45)
        self.b_log_var = nn.Parameter(log_var(n_out))
        self.q0_mean = nn.Parameter(small(n_out))
        self.q0_log_var = nn.Parameter(log_var(n_out))
        self.r0_c = nn.Parameter(small(n_out))
        self.r0_b1 = nn.Parameter(small(n_out))
        self.r0_b2 = nn.Parameter(small(n_out))
        self.flow_q = NormalizingFlow([RNVP(n_out, h_sizes=h_sizes) for _ in range(n_flows_q)])
        self.flow_r = NormalizingFlow([RNVP(n_out, h_sizes=h_sizes) for _ in range(n_flows_r)])

    def _apply(self, fn, *args, **kwargs):  # b_mean follows the module across devices
        super()._apply(fn, *args, **kwargs)
        self.b_mean = fn(self.b_mean)
        return self

    # ------------------------------------------------------------------ hot path (the flow)
    def sample_z(self, eps_z: Tensor | None = None, masks=None) -> tuple[Tensor, Tensor]:
        """(mnf_conv.py:90-98): z (1, n_out) and log|det J| (scalar) of flow_q at z0 = q0_mean + q0_std eps."""
        dev = self.q0_mean.device
        eps_z = torch.randn(1, self.n_out, device=dev) if eps_z is None else \
            eps_z.to(dev, torch.float32).reshape(1, self.n_out).contiguous()
        if dev.type != "cuda":
            raise RuntimeError(f"torch_mnf_amd: input must be a GPU tensor (got {dev.type}); the HIP path has no CPU fallback")
        z0 = _SampleZ0Fn.apply(self.q0_mean, self.q0_log_var, eps_z, self)
        z, log_det = _flow_through(self.flow_q, z0, masks)
        return z, log_det.squeeze()

    # ------------------------------------------------------------------ caller (stock PyTorch-ROCm)
    def forward(self, x: Tensor, eps: Tensor | None = None, eps_z: Tensor | None = None, masks=None) -> Tensor:
        """(mnf_conv.py:67-88, algorithm 2 of the paper)."""
        z, _ = self.sample_z(eps_z, masks)
        if not x.is_cuda or x.device != self.W_mean.device or x.dtype != torch.float32:
            raise RuntimeError(f"torch_mnf_amd: MNFConv2d.forward needs a float32 input on the layer's GPU "
                               f"({self.W_mean.device}), got {x.dtype} on {x.device}; the HIP path has no CPU fallback")
        # the convolutions' operands and the noise epilogue are one library launch each (and one each in backward); the
        # two convolutions are the caller's arithmetic (MIOpen)
        Wz, W_var, b_var = _ConvOperandsFn.apply(self.W_mean, self.W_log_var, self.b_log_var, z, self)
        b_mean = self.b_mean if self.b_mean.device == x.device else self.b_mean.to(x.device)
        mean = F.conv2d(x, weight=Wz, bias=b_mean)
        var = F.conv2d(x * x, weight=W_var, bias=b_var)
        return _NoiseFn.apply(mean, var, torch.randn_like(var) if eps is None else eps.to(var.device))

    def kl_div(self, noise: dict | None = None) -> Tensor:
        """(mnf_conv.py:100-133): both flows, then every closed-form term in one launch (``mnf_mnf_kl_fwd``).
        ``noise``: optional injected draws {"eps_z", "masks_q", "eps_w", "eps_b", "masks_r"}."""
        noise = noise or {}
        z, log_det_q = self.sample_z(noise.get("eps_z"), noise.get("masks_q"))
        z_r, log_det_r = _flow_through(self.flow_r, z, noise.get("masks_r"))
        dev = self.W_mean.device
        rows = self.W_mean.numel() // self.n_out
        eps_w, eps_b = noise.get("eps_w"), noise.get("eps_b")
        eps_w = torch.randn(rows, device=dev) if eps_w is None else eps_w.to(dev)
        eps_b = torch.randn(1, device=dev) if eps_b is None else eps_b.to(dev).reshape(1)
        if log_det_r.numel() != 1:  # the reference unpacks `[log_det_r]` (:127)
            raise ValueError(f"kl_div: flow_r returned {log_det_r.numel()} log-determinants for one row")
        b_mean = self.b_mean if self.b_mean.device == dev else self.b_mean.to(dev)
        return _MnfKlFn.apply(z, log_det_q, z_r, log_det_r, eps_w, eps_b, self, True, b_mean, self.W_mean,
                              self.W_log_var, self.b_log_var, self.q0_mean, self.q0_log_var, self.r0_c, self.r0_b1,
                              self.r0_b2)


class MNFLeNet(nn.Sequential):
    """The reference's Bayesian LeNet (models/mnf_lenet.py:8-33): the container the MNF layers are trained in
    (tests/test_mnf_mnist.py).  ``kl_div()`` is the sum over the MNF layers, as there."""

    def __init__(self, **kwargs) -> None:
        super().__init__(MNFConv2d(1, 20, kernel_size=5, **kwargs), nn.ReLU(), nn.MaxPool2d(kernel_size=2),
                         MNFConv2d(20, 50, kernel_size=5, **kwargs), nn.ReLU(), nn.MaxPool2d(kernel_size=2), nn.Flatten(),
                         MNFLinear(50 * 16, 50, **kwargs), nn.ReLU(), MNFLinear(50, 10, **kwargs), nn.LogSoftmax(dim=-1))

    def kl_div(self) -> Tensor:
        return sum(layer.kl_div() for layer in self if hasattr(layer, "kl_div"))


class MNFFeedForward(nn.Sequential):
    """The reference's MNF multilayer perceptron (models/mnf_feed_forward.py:14-41): MNFLinear / activation /
    BatchNorm1d blocks, the last activation and batch norm dropped."""

    def __init__(self, layer_sizes, activation=nn.ReLU, **kwargs) -> None:
        layers: list[nn.Module] = []
        for s1, s2 in zip(layer_sizes, layer_sizes[1:]):
            layers += [MNFLinear(s1, s2, **kwargs), activation(), nn.BatchNorm1d(s2)]
        super().__init__(*layers[:-2])

    def kl_div(self) -> Tensor:
        return sum(layer.kl_div() for layer in self if hasattr(layer, "kl_div"))

