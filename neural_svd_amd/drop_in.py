"""train_operator with the reference's signature (examples/operator/__init__.py:20-153) for
main_pde.py-style drivers: torch.optim.RMSprop + CosineAnnealingLR + EMA on ordinary nn.Parameters,
the loss / operator / eval math on the HIP path. Plotting and the local-energy monitor are left out
(they are not part of the hot path). For maximum step rate use trainer.FusedTrainer instead."""
from __future__ import annotations

import contextlib
import os
import time

import torch

from .spectrum import compute_spectrum_evd


class ExponentialMovingAverage:
    """torch_ema.ExponentialMovingAverage's behaviour as the reference uses it (update(),
    average_parameters(), state_dict()): shadow = clone(params); on update n += 1,
    d = min(decay, (1 + n) / (10 + n)), shadow -= (1 - d) (shadow - param)."""

    def __init__(self, parameters, decay: float, use_num_updates: bool = True):
        self.params = [p for p in parameters if p.requires_grad]
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in self.params]
        self.collected = None

    @torch.no_grad()
    def update(self):
        d = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            d = min(d, (1 + self.num_updates) / (10 + self.num_updates))
        diffs = torch._foreach_sub(self.shadow_params, [p.detach() for p in self.params])
        torch._foreach_mul_(diffs, 1.0 - d)
        torch._foreach_sub_(self.shadow_params, diffs)

    @contextlib.contextmanager
    def average_parameters(self):
        saved = [p.detach().clone() for p in self.params]
        with torch.no_grad():
            for p, s in zip(self.params, self.shadow_params):
                p.copy_(s)
        try:
            yield
        finally:
            with torch.no_grad():
                for p, s in zip(self.params, saved):
                    p.copy_(s)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)


def get_optimizer(args, model):
    """examples/utils.py:48-72 (rmsprop branch is what the PDE scripts use)."""
    if args.optimizer == "rmsprop":
        return torch.optim.RMSprop(model.parameters(), lr=args.lr, alpha=args.rmsprop_decay, eps=1e-10,
                                   weight_decay=0, momentum=args.momentum)
    if args.optimizer == "adam":
        return torch.optim.Adam(model.parameters(), lr=args.lr, eps=args.adam_eps)
    if args.optimizer == "sgd":
        return torch.optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum)
    raise NotImplementedError


def train_operator(args, method, operator, make_batch_ftn_train, val_data, batch_ftn_val, log_writer, log_file,
                   device, importance_train, importance_val, ground_truth_spectrum=None):
    optimizer = get_optimizer(args, method)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.num_iters)
    ema = ExponentialMovingAverage(method.parameters(), decay=args.ema_decay)
    all_eigvals, all_norms = [], []
    start = time.time()
    total_loss = 0.0
    for it in range(args.num_iters):
        method.train()
        optimizer.zero_grad()
        x = make_batch_ftn_train().to(device)
        x = x.reshape(x.shape[0], -1)
        loss, _aux = method.compute_loss_operator(operator, x, importance=importance_train)
        loss.backward()
        optimizer.step()
        if args.use_lr_scheduler:
            scheduler.step()
        ema.update()
        if (it + 1) % args.print_freq == 0:
            li = loss.item()  # the only host sync, and only at print time (the reference syncs every step)
            total_loss += li
            row = {"iter": it + 1, "train_loss": li, "avg_train_loss": total_loss / ((it + 1) // args.print_freq),
                   "time": time.time() - start}
            print(row)
            if log_writer is not None:
                log_writer.writerow(row)
                log_file.flush()
        if (it + 1) % args.eval_freq == 0:
            method.eval()
            with ema.average_parameters():
                if batch_ftn_val is not None:
                    outputs = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator,
                                                   importance_train=importance_train, importance_val=importance_val,
                                                   normalize=True, set_first_mode_const=False, device=device)
                    print(f"it{it + 1} eigvals: {outputs['eigvals']}")
                    print(f"it{it + 1} norms: {outputs['norms']}")
                    all_eigvals.append(outputs["eigvals"])
                    all_norms.append(outputs["norms"])
            if getattr(args, "log_dir", None):
                os.makedirs(args.log_dir, exist_ok=True)
                torch.save(dict(args=args, method=method.state_dict(), ema=ema.state_dict(),
                                optimizer=optimizer.state_dict()), os.path.join(args.log_dir, f"{it + 1}.pth"))
    return all_eigvals, all_norms
