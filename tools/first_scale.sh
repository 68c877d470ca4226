#!/bin/bash
# First measurement on a node with several MI355X: the training step and the sharded pool sweep at 1, 2, 4 and 8 GPUs, the
# bucketed gradient exchange under the backward pass (PACED_EARLY_UPDATE=1, default) against one exchange after it (=0).
# bench.py starts the rank processes itself (one per GPU, RCCL); every line carries "ranks": {"world_size", "backend"}
# as torch.distributed reports them, so a run that fell back to one rank cannot pass for a scaling point.
#   usage: bash tools/first_scale.sh [OUT_DIR] [GPU_COUNTS...]        default: gpurun_out/scale  1 2 4 8
set -o pipefail
O=${1:-gpurun_out/scale}; shift
COUNTS=${@:-1 2 4 8}
mkdir -p $O
for n in $COUNTS; do
  for early in 1 0; do
    [ $n = 1 ] && [ $early = 0 ] && continue
    PACED_EARLY_UPDATE=$early timeout -k 10 900 python bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --launch-timeout 800 \
        > $O/n${n}_early${early}.json 2> $O/n${n}_early${early}.err || { echo "n=$n early=$early failed"; tail -n 20 $O/n${n}_early${early}.err; exit 1; }
    python - $O/n${n}_early${early}.json $n <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); n = int(sys.argv[2])
assert d["ranks"]["world_size"] == n == d["n_gpus"], d["ranks"]
assert n == 1 or d["ranks"]["backend"] == "nccl", d["ranks"]
print(f'n={n} early_update={d["ranks"]["early_update"]}: {d["value"]:.1f} triples/s, {d["ms_per_step"]:.2f} ms/step, '
      f'pool sweep {d["secondary"]["ms_per_sweep"]:.1f} ms ({d["secondary"]["value"]:.3e} scores/s)')
PY
  done
done
