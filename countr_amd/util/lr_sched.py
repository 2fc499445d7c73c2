"""Per-iteration LR schedule: linear warm-up then half-cosine (reference util/lr_sched.py:9-21)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        span = args.epochs - args.warmup_epochs
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / span))
    if optimizer is not None:
        for group in optimizer.param_groups:
            group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
