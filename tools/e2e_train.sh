#!/bin/bash
# End-to-end check on a GPU box: synthetic Megapixel-MNIST in the reference's on-disk format -> ips_amd/main.py
# (sparse loader, patches built on the GPU, ips() + fused training step, evaluation), 3 epochs, with and without the HIP
# graph / the fused training kernels.  Prints the per-epoch log lines (loss, accuracy, images/s).
set -e
D=${1:-/tmp/mm_synth}
python -m ips_amd.main --make-synthetic $D --n-train 256 --n-test 64 > /dev/null
for extra in "" "--hip-graph" "--hip-graph --precision fp32x3"; do
  echo "== main.py $extra"
  python -m ips_amd.main --data-dir $D --patch 32 --stride 32 --M 64 --I 64 --B 16 --B-seq 16 --epochs 3 --workers 4 --sparse $extra 2>&1 | grep -o "^Train Epoch: [0-9]*\|^Test Epoch: [0-9]*\|avg. loss over tasks: [0-9.]*\|images_per_s: [0-9.]*" | paste - - - | tail -6
done
echo "== IPSX_TRAIN_FUSED=0 --hip-graph"
IPSX_TRAIN_FUSED=0 python -m ips_amd.main --data-dir $D --patch 32 --stride 32 --M 64 --I 64 --B 16 --B-seq 16 --epochs 3 --workers 4 --sparse --hip-graph 2>&1 | grep -o "^Train Epoch: [0-9]*\|^Test Epoch: [0-9]*\|avg. loss over tasks: [0-9.]*\|images_per_s: [0-9.]*" | paste - - - | tail -2
