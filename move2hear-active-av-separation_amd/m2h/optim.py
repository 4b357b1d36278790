"""Adam + gradient-norm clipping over flat buffers, arithmetic in libm2h.so (K19/K20 of SURVEY 2.2).

``FlatAdam`` is a ``torch.optim.Optimizer`` (so ``LambdaLR`` of ppo_trainer.py:711-718 drives its lr) whose parameters are views
into one flat fp32 buffer; the gradients autograd leaves in ``p.grad`` are gathered into a second flat buffer by one batched
copy (``m2h_rows_copy``: zero_grad() sets the grads to None, so autograd hands its gradient tensors over instead of launching one
accumulation add per parameter into pre-zeroed storage -- 32 adds and a memset per policy epoch): one sum-of-squares reduction
then gives the clip coefficient (kept on the device, no host sync), one kernel applies clip + Adam to all parameters, and -- for
DD-PPO -- the flat gradient buffer is the single RCCL all-reduce payload (23.3 MB for the policy).  Semantics are torch.optim.Adam's (betas 0.9/0.999, no weight
decay, no amsgrad) with eps from the config (1e-5, ppo.py:48-55) and nn.utils.clip_grad_norm_'s coefficient.
"""
import torch

from . import _lib, ops


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999)):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, eps=eps, betas=betas))
        self._built = False
        self.t = 0

    def _build(self):
        """Flat buffers over the parameters that require grad NOW (the trainer may freeze modules after construction)."""
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not ps:
            raise ValueError("FlatAdam: no trainable parameters")
        dev = ps[0].device
        if not ps[0].is_cuda:
            raise RuntimeError("FlatAdam: parameters must live on the GPU (no CPU path)")
        n = sum(p.numel() for p in ps)
        self._ps = ps
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.coef = torch.ones(2, device=dev, dtype=torch.float32)
        self._coef_dirty = False
        self._hyper = torch.zeros(4, device=dev, dtype=torch.float32)    # lr, 1 - beta1^t, sqrt(1 - beta2^t) of captured_step launches
        self._scratch = torch.empty(1024, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)          # parameter storage now lives in the flat buffer
                off += k
        self.n = n
        self._offsets = []
        off = 0
        for p in ps:
            self._offsets.append(off)
            off += p.numel()
        self._no_idx = torch.zeros(1, device=dev, dtype=torch.int64)      # rows_copy's index argument (no item uses an index)
        self._gathered = True                                             # flat_g (zeros) is consistent with "no gradients yet"
        self._gathered_idx = set()                                        # parameters gathered since the last step (partial gathers: grad_bucket)
        self._slots_used = set()                                          # offsets handed out by functional.grad_slot since the last zero_grad
        self._built = True
        from . import functional
        functional._flat_optimizers.add(self)

    def build(self):
        """Moves the trainable parameters into the flat buffer now (otherwise done by the first zero_grad).  Anything that
        holds parameter addresses across steps -- the trainer's HIP graphs -- calls this first."""
        if not self._built:
            self._build()

    def zero_grad(self, set_to_none=True):
        """Gradients to None (autograd then hands over its gradient tensors; they are gathered into the flat buffer by
        grad_buffer() / step()).  set_to_none=False zeroes existing gradient tensors in place instead."""
        if not self._built:
            self._build()
        for p in self._ps:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
        self._gathered_idx = set()
        if set_to_none:
            self._slots_used = set()     # every gradient of the coming backward may be written straight into its slice (functional.grad_slot)
        self._gathered = False

    @torch.no_grad()
    def _gather(self, idx=None):
        """p.grad of every parameter (idx: of those parameters only) -> its slice of the flat gradient buffer (one batched copy; None
        counts as zeros).  A parameter gathered once is not gathered again before the next step / zero_grad: its slice may hold an
        all-reduced sum by then (bucketed reduction, ddppo_utils.GradReduceStep.early)."""
        from . import functional
        functional.join_wgrad_branches(self.flat_g.device)   # weight gradients forked to a side branch of a captured step (functional.wgrad_side_branches)
        if self._gathered:
            return
        done = self._gathered_idx
        todo = [i for i in (range(len(self._ps)) if idx is None else idx) if i not in done]
        items, missing = [], []
        for i in todo:
            p, off = self._ps[i], self._offsets[i]
            g = p.grad
            if g is None:
                missing.append((off, p.numel()))
                continue
            if g.dtype != torch.float32 or g.device != self.flat_g.device:
                raise RuntimeError("FlatAdam: gradients must be fp32 tensors on the parameters' GPU")
            dst = self.flat_g[off:off + p.numel()]
            if g.data_ptr() == dst.data_ptr():
                continue                                   # already a view of its slice (functional.grad_slot)
            items.append((g.contiguous().view(-1), dst, -1, -1))
        if missing and len(missing) == len(self._ps):
            self.flat_g.zero_()
        else:
            for off, k in missing:                         # (gradients that live in the buffer must survive: zero the absent ones' slices only)
                self.flat_g[off:off + k].zero_()
        if items:
            ops.rows_copy(items, self._no_idx)
        done.update(todo)
        if len(done) == len(self._ps):
            self._gathered = True
            self._gathered_idx = set()

    def grad_buffer(self):
        """The flat gradient (all-reduce payload), gathered from the parameters' .grad."""
        if not self._built:
            self._build()
        self._gather()
        return self.flat_g

    def grad_bucket(self, params):
        """The slice of the flat gradient that belongs to `params` (a contiguous run of this optimizer's parameters), gathered from
        their .grad now -- a bucket whose backward is complete while the rest of the backward is still being enqueued."""
        b0, b1, idx = self.param_range(params)
        self._gather(idx)
        return self.flat_g[b0:b1]

    @torch.no_grad()
    def step(self, max_grad_norm=None, grad_scale=1.0):
        """clip_grad_norm_(params, max_grad_norm) followed by Adam.step(); grad_scale multiplies the gradient first
        (1/world_size after a sum all-reduce; the clip norm is computed on the scaled gradient like DDP's averaged grads)."""
        if not self._built:
            raise RuntimeError("FlatAdam.step before zero_grad()/backward")
        self._gather()
        self._gathered = False   # the next step gathers afresh (gradients may be replaced without zero_grad: HIP-graph replays)
        self._gathered_idx = set()
        g = self.param_groups[0]
        lr, eps, (b1, b2) = g["lr"], g["eps"], g["betas"]
        self.t += 1
        lib = _lib.load()
        dev = self.flat_p.device
        with torch.cuda.device(dev):
            st = ops._stream(self.flat_p)
            if grad_scale != 1.0:
                self.flat_g.mul_(grad_scale)
            if max_grad_norm is None and not self.measure_grad_norm:
                # no clipping asked for (passive pre-training: the reference's clip_grad_norm_ acts on zeroed gradients)
                # and nobody reads the norm: the two reduction launches over the whole gradient are skipped, the factor stays 1
                if self._coef_dirty:
                    self.coef.fill_(1.0)
                    self._coef_dirty = False
            else:
                mg = float(max_grad_norm) if max_grad_norm is not None else 0.0
                _lib.check(lib.m2h_grad_clip_coef(ops._ptr(self.flat_g), self.n, mg, ops._ptr(self.coef), ops._ptr(self._scratch), st),
                           "m2h_grad_clip_coef")
                self._coef_dirty = True
            _lib.check(lib.m2h_adam_step(ops._ptr(self.flat_p), ops._ptr(self.flat_g), ops._ptr(self.exp_avg), ops._ptr(self.exp_avg_sq),
                                         self.n, float(lr), float(b1), float(b2), float(eps), self.t, ops._ptr(self.coef), 1.0, st),
                       "m2h_adam_step")
        from . import functional
        functional.bump_param_epoch()

    # ------------------------------------------------------------------ the step inside a captured HIP graph
    def param_range(self, params):
        """[begin, end) of `params` in the flat buffers; they must be a contiguous run of this optimizer's parameters."""
        if not self._built:
            self._build()
        ids = {id(p) for p in params if p.requires_grad}
        idx = [i for i, p in enumerate(self._ps) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)) or len(idx) != len(ids):
            raise ValueError("FlatAdam.param_range: the parameters are not a contiguous run of this optimizer's")
        return self._offsets[idx[0]], self._offsets[idx[-1]] + self._ps[idx[-1]].numel(), idx

    @torch.no_grad()
    def captured_step(self, params=None):
        """Adam on the current gradients of `params` (default: all), to be CAPTURED into a HIP graph right behind the backward that
        produced them (no clipping: passive pre-training).  The step count and the learning rate reach the kernel through
        a device buffer: call ``begin_replayed_step()`` on the host before every replay (once per step, however many ranges the graph
        updates), ``end_replayed_step()`` after it."""
        if len(self.param_groups) != 1:
            raise RuntimeError("FlatAdam.captured_step: one parameter group only (the captured step reads param_groups[0]'s lr / betas / eps)")
        b0, b1_, idx = self.param_range(params if params is not None else self._ps)
        from . import functional
        functional.join_wgrad_branches(self.flat_g.device, forked_from=torch.cuda.current_stream(self.flat_g.device))   # (another branch's are its own step's business)
        items = []
        for i in idx:
            p, off = self._ps[i], self._offsets[i]
            if p.grad is None:
                raise RuntimeError("FlatAdam.captured_step: a parameter of the range has no gradient")
            dst = self.flat_g[off:off + p.numel()]
            if p.grad.data_ptr() != dst.data_ptr():
                items.append((p.grad.contiguous().view(-1), dst, -1, -1))
        if items:
            ops.rows_copy(items, self._no_idx)
        g = self.param_groups[0]
        eps, (b1, b2) = g["eps"], g["betas"]
        lib = _lib.load()
        sl = slice(b0, b1_)
        with torch.cuda.device(self.flat_p.device):
            _lib.check(lib.m2h_adam_step_dev(ops._ptr(self.flat_p[sl]), ops._ptr(self.flat_g[sl]), ops._ptr(self.exp_avg[sl]),
                                             ops._ptr(self.exp_avg_sq[sl]), b1_ - b0, ops._ptr(self._hyper), float(b1), float(b2), float(eps),
                                             None, 1.0, ops._stream(self.flat_p)), "m2h_adam_step_dev")

    def begin_replayed_step(self):
        """Host side of a replayed step whose graph holds ``captured_step`` launches: count the step, upload lr and the bias corrections."""
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.t += 1
        with torch.cuda.device(self.flat_p.device):
            _lib.check(_lib.load().m2h_adam_hyper(float(g["lr"]), float(b1), float(b2), self.t, ops._ptr(self._hyper), ops._stream(self._hyper)),
                       "m2h_adam_hyper")

    def end_replayed_step(self):
        from . import functional
        self._gathered = False
        functional.bump_param_epoch()

    measure_grad_norm = False   # True: step(max_grad_norm=None) still measures ||g||_2 (grad_norm()); off: those two launches are skipped

    def grad_norm(self):
        """||g||_2 measured by the last step that clipped or had ``measure_grad_norm`` set (device tensor)."""
        return self.coef[1]
