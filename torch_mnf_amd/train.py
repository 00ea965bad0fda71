"""Training support for the flow path: the model's parameters in ONE buffer, the optimizer in ONE launch.

The reference trains its flows with ``torch.optim.Adam(model.parameters())`` (tests/test_flows.py:41-50,
examples/half_moons.ipynb:170-200).  That works unchanged on these modules; but a 9-layer AffineHalfFlow stack has
144 small parameter tensors, and at the batch sizes the reference trains at a step is then bound by the host's
handling of 144 tensors (concatenate for the kernels, scatter the gradient back, a 144-entry optimizer list), not by
the GPU.  ``FlatParameters`` re-homes every parameter of a model as a view of one contiguous fp32 buffer (names,
shapes and ``state_dict`` keys unchanged) with every gradient a view of a second one; the fused layers then read
their parameter slice and write their gradient slice in place, and ``FusedAdam`` updates the whole buffer with one
``mnf_adam_step`` launch.

    flat = FlatParameters(model)
    opt = FusedAdam(flat, lr=1e-3)
    for x in batches:
        opt.zero_grad()                      # one memset
        loss = -model.log_prob(x).mean()
        loss.backward()
        opt.step()                           # one launch
"""
from __future__ import annotations

import functools

import torch
from torch import Tensor, nn

from . import _lib
from .flows import _stream

__all__ = ["FlatParameters", "FusedAdam", "GraphedStep"]


def _flat_zero_grad(model, set_to_none: bool = True) -> None:  # noqa: ARG001 (nn.Module.zero_grad's signature)
    """``model.zero_grad()`` of a model whose parameters live in a FlatParameters buffer: the buffer's one memset."""
    flat = model.__dict__.get("_mnf_flat")
    if flat is None:
        return nn.Module.zero_grad(model, set_to_none)
    flat.zero_grad()


class FlatParameters:
    """Every trainable parameter of ``model`` as a view of ``self.data`` (fp32, one device), every gradient a view
    of ``self.grad``, in ``model.parameters()`` order.  ``state_dict`` / ``load_state_dict`` keep working (they copy
    into the views).  ``generation`` is bumped by whoever rewrites ``data`` in place (FusedAdam.step,
    ``touch()``): the fused layers key their packed operand images on it."""

    def __init__(self, model: nn.Module) -> None:
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("the model has no trainable parameters")
        device = params[0].device
        if any(p.device != device or p.dtype != torch.float32 for p in params):
            raise ValueError("FlatParameters needs every parameter in float32 on one device")
        n = sum(p.numel() for p in params)
        self.model = model
        self.data = torch.empty(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.offset: dict[int, int] = {}
        self.generation = 0
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.data[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.data[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                self.offset[id(p)] = off
                off += k
        self.params = params
        self._grad_views = [p.grad for p in params]  # identity-compared in _reattach (one `is` per parameter)
        for m in model.modules():  # the fused layers look here for their slices
            m.__dict__["_mnf_flat"] = self
        # The reference's loops clear gradients with ``model.zero_grad()`` (tests/test_flows.py:27,
        # examples/half_moons.ipynb:188), whose default sets every ``p.grad`` to None: the next backward would then
        # give each parameter a fresh gradient tensor outside ``self.grad``, and the fused layers' in-place sums in
        # ``self.grad`` would never be cleared.  On this model ``zero_grad`` is the one memset instead.
        # (a module-level function bound to the model with functools.partial and resolved through the module at call
        # time: picklable, and a copy.deepcopy of the model clears ITS OWN buffer, not this one's)
        model.zero_grad = functools.partial(_flat_zero_grad, model)
        if hasattr(model, "invalidate"):
            model.invalidate()

    def slice_of(self, params: list[Tensor]) -> tuple[int, int] | None:
        """(offset, length) of ``params`` inside the buffers if they sit there back to back in this order."""
        off0 = self.offset.get(id(params[0]))
        if off0 is None:
            return None
        off = off0
        for p in params:
            if self.offset.get(id(p)) != off:
                return None
            off += p.numel()
        return off0, off - off0

    def _reattach(self, fold: bool) -> None:
        """Make every ``p.grad`` the view of ``self.grad`` again.  A torch optimizer's ``zero_grad(set_to_none=True)``
        (or ``p.grad = None``) detaches a parameter from the buffer: autograd then allocates a stray gradient tensor
        that FusedAdam would never see.  ``fold``: add a stray gradient's content into the view first (called before
        an optimizer step: nothing a backward pass produced may be lost).  One identity comparison per parameter --
        ``p.grad`` returns the same Python object as long as nobody replaced it."""
        for i, p in enumerate(self.params):
            g = p.grad
            view = self._grad_views[i]
            if g is view:
                continue
            if fold and g is not None:
                view.add_(g.detach().to(view.dtype).view(view.shape))
            p.grad = view

    def home_is_valid(self, params: list[Tensor]) -> bool:
        """Do ``params`` still live in ``self.data`` (every data pointer where the offsets say) and do they
        all still want gradients?  ``model.to()``, ``.float()``, ``load_state_dict(assign=True)`` re-home parameters
        behind this object's back, and ``requires_grad_(False)`` freezes one; a fused layer that wrote its gradient
        slice in place would then update the wrong memory, or a frozen parameter."""
        base = self.data.data_ptr()
        for p in params:  # every parameter of the list (<= ~9 of them): a middle one may have been re-homed or frozen
            off = self.offset.get(id(p))
            if off is None or p.data_ptr() != base + 4 * off or not p.requires_grad:
                return False
        return True

    def zero_grad(self) -> None:
        """One memset; the per-parameter ``.grad`` views stay in place (re-attached if someone set them to None or
        let autograd allocate a stray gradient: ``set_to_none`` semantics would detach them from the buffer)."""
        self.grad.zero_()
        self._reattach(fold=False)

    def touch(self) -> None:
        """Call after writing ``data`` (or a parameter's ``.data``) in place by other means than FusedAdam."""
        self.generation += 1


class FusedAdam:
    """``torch.optim.Adam`` semantics (no amsgrad; ``weight_decay`` as L2 on the gradient) on a FlatParameters
    buffer: one ``mnf_adam_step`` launch per step."""

    def __init__(self, flat: FlatParameters, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, capturable: bool = False) -> None:
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), \
            float(weight_decay)
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.steps = 0
        # capturable: the step counter (and the two bias-correction factors) live on the device, so that a step
        # captured in a hipGraph (GraphedStep) advances them on every replay
        self.state = torch.zeros(3, dtype=torch.float32, device=flat.data.device) if capturable else None

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self) -> None:
        f = self.flat
        f._reattach(fold=True)  # a gradient that landed outside the buffer (p.grad was None at backward) is added in
        self.steps += 1
        if self.state is not None:
            _lib.check("mnf_adam_step_graph", _lib.load().mnf_adam_step_graph(
                f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                f.data.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                self.state.data_ptr(), _stream()))
            f.generation += 1
            return
        _lib.check("mnf_adam_step", _lib.load().mnf_adam_step(
            f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), f.data.numel(),
            self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps, _stream()))
        f.generation += 1


class GraphedStep:
    """One training step -- zero_grad, ``loss_fn(*batch)``, backward, ``opt.step()`` -- captured once in a hipGraph and
    replayed per batch: at the batch sizes the reference trains at (128 rows: examples/half_moons.ipynb:170-200, the
    MNF MNIST example) a step is dozens to a thousand kernels of microseconds each, and what it costs is their
    launches.

        opt = FusedAdam(FlatParameters(model), lr=1e-3, capturable=True)
        step = GraphedStep(opt, lambda x: -model.log_prob(x).mean(), example_batch)
        for x in batches:                 # every batch with example_batch's shape
            loss = step(x)                # device tensor, rewritten by the next call

    ``opt`` is a ``FusedAdam(..., capturable=True)`` or any torch optimizer built with ``capturable=True`` (pass
    ``model=`` then: its layers' packed operand images are invalidated after every replay); ``example`` is a tensor or
    a tuple of tensors (inputs, labels, ...), and the step is called with batches of exactly those shapes.
    The captured kernels are the ones an eager step launches (the Flow modules' HIP kernels run on the capture stream;
    parameters, gradients and optimizer state are updated in place by a replay; the operand images are repacked
    inside the graph).  Everything the step decides on the host is frozen at capture time: batch shapes, kernel
    choices, host-drawn random numbers.  RNVP layers therefore draw their masks on the device while the step is
    warmed up and recorded (``flows.device_drawn_masks``: torch.bernoulli, the reference's own draw, redrawn by every
    replay like torch.randn is) instead of hashing a host-drawn seed in the kernel, and ``MNFLinear.forward`` draws its
    output noise with torch.randn instead of seeding the in-kernel generator from the host, for the same reason.  Glow captures too: its
    permutation sits on the device once, and its training-path inverse is torch.linalg.inv_ex (no host-side
    singularity check).  If eager steps ran before, drop their loss tensors first (``del loss``): a live loss keeps that step's autograd
    graph, and with it gradient-accumulation nodes bound to the default stream, alive into the capture."""

    def __init__(self, opt, loss_fn, example, warmup: int = 3, model: nn.Module | None = None) -> None:
        from .flows import device_drawn_masks

        fused = isinstance(opt, FusedAdam)
        if fused and opt.state is None:
            raise ValueError("GraphedStep needs FusedAdam(..., capturable=True): the step counter must live on the device")
        if not fused and any(not g.get("capturable", False) for g in opt.param_groups):
            raise ValueError("GraphedStep needs a torch optimizer built with capturable=True (its step counters must "
                             "live on the device)")
        batch = tuple(example) if isinstance(example, (tuple, list)) else (example,)
        if not batch or not all(isinstance(t, Tensor) and t.is_cuda for t in batch):
            raise ValueError("GraphedStep captures a hipGraph: the example batch must be device tensor(s)")
        self.opt, self.loss_fn, self.fused = opt, loss_fn, fused
        self.model = model if model is not None else (opt.flat.model if fused else None)
        self.batch = tuple(t.detach().clone() for t in batch)
        device = self.batch[0].device
        # every layer that caches packed operand images per parameter version
        self._invalidators = [] if self.model is None else [
            m.invalidate for m in self.model.modules() if callable(getattr(m, "invalidate", None))]

        def run():
            if fused:
                opt.zero_grad()  # (one memset of the flat gradient buffer, inside the graph)
            loss = loss_fn(*self.batch)
            loss.backward()
            opt.step()
            return loss

        # warm-up on a side stream (torch.cuda.graphs' recipe): one-time attribute calls, index tables, algorithm
        # searches and caches happen here, not under capture.  These are real steps on the example batch.
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), device_drawn_masks():
            for _ in range(warmup):
                if not fused:
                    opt.zero_grad(set_to_none=True)
                run()
        torch.cuda.current_stream(device).wait_stream(side)
        if not fused:
            opt.zero_grad(set_to_none=True)  # the gradients then live in the graph's memory pool, rewritten per replay
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph), device_drawn_masks():
                self.loss = run().detach()
        except RuntimeError as err:  # (torch.AcceleratorError is a RuntimeError)
            raise RuntimeError(
                "GraphedStep: the step contains an operation that cannot be recorded in a hipGraph (a host "
                "synchronisation, a host-to-device copy of pageable memory, a .item()); the chained exception "
                "names the call") from err
        if fused:
            opt.steps -= 1           # (capture records the step, it does not run it)
        self._touch()                # operand images "packed" under capture were only recorded: eager code repacks
        self.replays = 0

    def _touch(self) -> None:
        """The parameters changed behind Python's back (a replay bumps no version counter): whatever the layers cache
        per parameter version must be rebuilt by the next eager pass."""
        if self.fused:
            self.opt.flat.generation += 1  # (the layers key their images on this counter)
            return
        for inv in self._invalidators:
            inv()

    def __call__(self, *batch: Tensor) -> Tensor:
        if len(batch) != len(self.batch) or any(b.shape != s.shape for b, s in zip(batch, self.batch)):
            raise ValueError(f"GraphedStep was captured for batches of shapes {[tuple(s.shape) for s in self.batch]}, "
                             f"got {[tuple(b.shape) for b in batch]}")
        for s, b in zip(self.batch, batch):
            s.copy_(b)
        self.graph.replay()
        self.replays += 1
        if self.fused:
            self.opt.steps += 1
        self._touch()
        return self.loss
