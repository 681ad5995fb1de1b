"""create_optimizer(args, model) for the MI355X engine (reference: optim_factory.py:57-187).

Only `--opt adamw` (the optimizer the pre-training recipe uses, README.md:53-78) is provided; it is the reference's
custom_optim.AdamW (custom_optim/adamw.py:55-121, _functional.py:115-140) as ONE fused kernel over the flat
parameter arena instead of a per-tensor Python loop: p, grad, exp_avg, exp_avg_sq are four flat fp32 buffers, the
decay / no-decay split of get_parameter_groups (optim_factory.py:57-100) is a per-1KiB-granule flag table.
`param_groups` keeps the reference's shape ({lr, weight_decay, lr_scale, betas, eps, params}) because the engine
rewrites lr / weight_decay every step (engine_for_pretraining_moco.py:60-66)."""
import json

import torch

import os

from . import ops

# the optimizer launch also writes the bf16 operand shadow and the transposed weight copies the next forward needs (dig_adamw_step_tr);
# "0": the plain launch, the forward re-casts and re-transposes every step (the round-5 plan)
FOLD_SHADOW = os.environ.get("DIG_ADAMW_FOLD", "1") != "0"


def get_parameter_groups(model, weight_decay=1e-5, skip_list=()):
    """Names per group, same rule as optim_factory.py:63-69: 1-D tensors, *.bias and skip-listed names do not decay."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(name)
        else:
            decay.append(name)
    return decay, no_decay


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, skip_list=()):
        self.model = model
        decay, no_decay = get_parameter_groups(model, weight_decay, skip_list)
        named = dict(model.named_parameters())
        # the arena's flag table was laid out with the same rule; verify instead of trusting
        for n in decay:
            assert model.specs[n].group in (0, 2), n          # (2: a parameter the model never reads -- no gradient, left untouched)
        for n in no_decay:
            assert model.specs[n].group == 1, n
        groups = [
            {"params": [named[n] for n in decay], "weight_decay": weight_decay, "lr_scale": 1.0, "names": decay},
            {"params": [named[n] for n in no_decay], "weight_decay": 0.0, "lr_scale": 1.0, "names": no_decay},
        ]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._bind()

    def _bind(self):
        dev = self.model.flat_params.device
        if getattr(self, "exp_avg", None) is None or self.exp_avg.device != dev:
            old = getattr(self, "exp_avg", None)
            self.exp_avg = torch.zeros_like(self.model.flat_params) if old is None else old.to(dev)
            old = getattr(self, "exp_avg_sq", None)
            self.exp_avg_sq = torch.zeros_like(self.model.flat_params) if old is None else old.to(dev)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are views of one arena that the backward kernels accumulate into: zero it with one kernel."""
        g = self.model.flat_grads
        if g.is_cuda:
            ops.fill_f32(g, 0.0)
        else:
            g.zero_()
        # The next backward of the model's own step may WRITE the single-contribution gradients of the BatchNorm-MLP heads instead of adding to
        # zeros (engine_core: one launch less per head layer).  CONTRACT: nothing else may add into those .grad views between this call and
        # that backward (an extra autograd path through the head weights, a regulariser, a hook) -- it would be overwritten; call
        # `model._grads_fresh = False` after zero_grad() to get torch's accumulate-always behaviour.
        self.model._grads_fresh = True

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, finite_gate=None, dev_scalars=None):
        """finite_gate: device float (the squared gradient norm); a non-finite value turns the launch into a no-op.
        dev_scalars: device fp32 [lr0, wd0, lr1, wd1, 1/bc1, 1/sqrt(bc2)] of THIS step (step_graph.adamw_scalars): the launch reads
        them at run time and the caller advances `_step` (a captured launch is replayed, this method is not)."""
        self._bind()
        g0, g1 = self.param_groups
        b1, b2 = g0["betas"]
        M = self.model
        if dev_scalars is not None:
            ops.adamw_step_dev(M.flat_params, M.flat_grads, self.exp_avg, self.exp_avg_sq, None, M.flat_groups, dev_scalars, b1, b2,
                               g0["eps"], grad_scale, finite_gate)
            if hasattr(M, "mark_weights_changed"):
                M.mark_weights_changed()
            return
        self._step += 1
        tr = M.transposed_weight_table() if (FOLD_SHADOW and hasattr(M, "transposed_weight_table") and M.flat_params.is_cuda
                                             and not torch.cuda.is_current_stream_capturing()) else None
        if tr is not None:
            # one launch: the update, the bf16 operand shadow of the whole arena and the transposed copies of the MLP / projection weights
            # (what the next forward would otherwise rebuild with a cast launch and three transpose launches)
            table, n_mats, n_tiles, tr_out, _, flags = tr[:6]
            ops.adamw_step_tr(M.flat_params, M.flat_grads, self.exp_avg, self.exp_avg_sq, M.shadow("online"), flags,
                              g0["lr"], g0["weight_decay"], g1["lr"], g1["weight_decay"], b1, b2, g0["eps"], self._step, table, n_mats, n_tiles,
                              tr_out, grad_scale, finite_gate)
            M._set_fresh()
            return
        ops.adamw_step(M.flat_params, M.flat_grads, self.exp_avg, self.exp_avg_sq, None, M.flat_groups,
                       g0["lr"], g0["weight_decay"], g1["lr"], g1["weight_decay"], b1, b2, g0["eps"], self._step, grad_scale, finite_gate)
        if hasattr(M, "mark_weights_changed"):
            M.mark_weights_changed()

    # ---- checkpoint format: the reference's torch-Optimizer layout (custom_optim/optimizer.py state_dict):
    #   {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [i, ...]}, ...]}
    # with i enumerating the parameters group by group (decay first), so checkpoints interchange with the reference.
    def _named_specs(self):
        out = []
        for g in self.param_groups:
            out.extend(g["names"])
        return out

    def state_dict(self):
        names = self._named_specs()
        state = {}
        if self._step > 0:
            for i, n in enumerate(names):
                sp = self.model.specs[n]
                state[i] = {"step": self._step,
                            "exp_avg": self.exp_avg[sp.offset:sp.offset + sp.numel].view(sp.shape),
                            "exp_avg_sq": self.exp_avg_sq[sp.offset:sp.offset + sp.numel].view(sp.shape)}
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: v for key, v in g.items() if key not in ("params", "names")}
            d.setdefault("amsgrad", False)
            d["params"] = list(range(k, k + len(g["names"])))
            k += len(g["names"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._bind()
        names = self._named_specs()
        if "state" not in sd:                                   # round-0 flat format of this package
            self._step = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        if len(sd["param_groups"]) != len(self.param_groups) or \
                [len(g["params"]) for g in sd["param_groups"]] != [len(g["names"]) for g in self.param_groups]:
            raise ValueError("loaded state dict has different parameter groups")
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            sp = self.model.specs[names[int(i)]]
            if tuple(st["exp_avg"].shape) != sp.shape:
                raise ValueError(f"optimizer state {i} ({names[int(i)]}): shape {tuple(st['exp_avg'].shape)} != {sp.shape}")
            self.exp_avg[sp.offset:sp.offset + sp.numel].view(sp.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[sp.offset:sp.offset + sp.numel].view(sp.shape).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused optimizer keeps one step counter")
        self._step = steps.pop() if steps else 0
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s_.items() if k not in ("params", "names", "amsgrad")})


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    opt_lower = args.opt.lower()
    if opt_lower.split('_')[-1] != 'adamw':
        raise NotImplementedError("dig_amd provides the fused AdamW of the pre-training recipe only (--opt adamw)")
    if get_num_layer is not None or get_layer_scale is not None:
        raise NotImplementedError("layer-wise lr decay belongs to the fine-tune path")
    module = model.module if hasattr(model, "module") else model
    skip = skip_list if skip_list is not None else (module.no_weight_decay() if hasattr(module, "no_weight_decay") else ())
    kw = dict(lr=args.lr, weight_decay=args.weight_decay if filter_bias_and_bn else 0.0)
    if getattr(args, "opt_eps", None) is not None:
        kw["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        kw["betas"] = tuple(args.opt_betas)
    print("optimizer settings:", kw)
    opt = FusedAdamW(module, skip_list=skip, **kw)
    print("Param groups = %s" % json.dumps({("decay" if i == 0 else "no_decay"): {"weight_decay": g["weight_decay"], "n_params": len(g["names"]), "lr_scale": g["lr_scale"]} for i, g in enumerate(opt.param_groups)}))
    return opt
