"""CPU restatement of the reference's optimiser side (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package).

What it restates, and what pins it
  * parameter grouping            optim.py:23-69 (create_optimizer), optim.py:4-21 (create_L0_optimizer): PINNED by
                                  tests/golden/optim_groups.json, captured from the reference's own functions
                                  (oracle/gen_golden.py gen_optim).
  * linear LR schedule            scheduler.py:14-22 (LambdaLR factor): restated, trivially checkable.
  * gradient clipping             apex_ddp_accelerator.py:99-102 = torch.nn.utils.clip_grad_norm_ (2-norm over all
                                  gradients, coefficient max_norm / (norm + 1e-6) clamped to 1): checked against torch.
  * AdamW step                    `transformers.optimization.AdamW` of transformers==4.12.5 (requirements.txt:2), the
                                  class optim.py:1 imports.  THAT PACKAGE VERSION IS NOT IN THIS IMAGE (transformers 5.x
                                  dropped the class), so its published algorithm is restated below and the arithmetic
                                  is "parity unpinned": no captured vectors exist for it.  It is cross-checked against
                                  torch.optim.AdamW, which differs only in where eps enters the denominator and in
                                  applying the decay before instead of after the Adam update (both O(eps) / O(lr^2*wd)).

transformers 4.12.5, AdamW.step (correct_bias=True, the default the reference uses):
    exp_avg    = beta1 * exp_avg    + (1 - beta1) * grad
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad^2
    denom      = sqrt(exp_avg_sq) + eps
    step_size  = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    p          = p - step_size * exp_avg / denom
    p          = p - lr * weight_decay * p          (only if weight_decay > 0; uses the UPDATED p)
"""
import math

import torch

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight")          # optim.py:35-43 (substring match on the parameter name)


def param_groups(named_params, init_params, lr, weight_decay, lr_mult=1):
    """optim.py:23-69: four groups (decay, no-decay) x (lr, lr*lr_mult for names in model.init_params), in
    named_parameters() order; frozen parameters skipped.  Returns [{lr, weight_decay, names}]."""
    groups = [dict(lr=lr, weight_decay=weight_decay, names=[]), dict(lr=lr, weight_decay=0.0, names=[]),
              dict(lr=lr * lr_mult, weight_decay=weight_decay, names=[]), dict(lr=lr * lr_mult, weight_decay=0.0, names=[])]
    large = set(init_params or [])
    for n, p in named_params:
        if not p.requires_grad:
            continue
        nd = any(s in n for s in NO_DECAY)
        groups[(3 if n in large else 1) if nd else (2 if n in large else 0)]["names"].append(n)
    return groups


def l0_param_groups(l0_named_params, reg_learning_rate):
    """optim.py:4-21: gate parameters descend with +reg_lr, the two Lagrange multipliers ASCEND (lr = -reg_lr);
    no weight decay, betas (0.9, 0.98), eps 1e-8."""
    named = list(l0_named_params)
    return ([dict(lr=reg_learning_rate, weight_decay=0.0, names=[n for n, _ in named if "lambda" not in n])],
            [dict(lr=-reg_learning_rate, weight_decay=0.0, names=[n for n, _ in named if "lambda" in n])])


def linear_schedule(step, num_warmup_steps, num_training_steps):
    """scheduler.py:14-22"""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (apex_ddp_accelerator.py:99-102): in-place; returns the total norm"""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def hf_adamw_step(p, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0):
    """one transformers-4.12.5 AdamW update of one tensor, in place (fp32); `step` counts from 1"""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(grad, alpha=1.0 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    denom = exp_avg_sq.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(exp_avg, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)
    return p
