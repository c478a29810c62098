#!/bin/bash
# Runs ON THE GPU BOX: what accounts for the gap between configs[2]'s end-of-training eigenvalue error here (1.35e-3)
# and the reference's published oscillator figures? One factor at a time, 100 000 steps each (~2.5 min):
#   stencil training, seeds 0-2, evaluated with the stencil AND with the exact Laplacian   (evaluation noise, seed spread)
#   exact-Laplacian training                                                                  (training-time FD noise)
#   joint instead of sequential nesting                                                       (nesting order)
#   the reference API end to end (host torch.randn sampler, compute_spectrum_evd)             (sampler, evaluation code)
out=/root/repo/gpurun_out/${1:-oscgap}
mkdir -p $out
cd /root/repo
for s in 0 1 2; do
  python scripts/train_hydrogen.py --problem oscillator --seed $s --evals 100000 --eval-exact --out $out/stencil_seed$s.json > $out/stencil_seed$s.log 2>&1
done
python scripts/train_hydrogen.py --problem oscillator --seed 0 --evals 100000 --laplacian-eps 0 --out $out/exact_seed0.json > $out/exact_seed0.log 2>&1
python scripts/train_hydrogen_dropin.py --problem oscillator --sequential --steps 100000 --eval-freq 100000 --seed 0 --out $out/dropin_seed0.json > $out/dropin_seed0.log 2>&1
python scripts/train_hydrogen_dropin.py --problem oscillator --steps 100000 --eval-freq 100000 --seed 0 --out $out/dropin_joint_seed0.json > $out/dropin_joint_seed0.log 2>&1
tail -n 2 $out/*.log
