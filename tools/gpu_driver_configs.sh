#!/bin/bash
# BASELINE.json configs[1..3] through the C++ driver on ONE GPU (reference-compatible command line, loop on the GPU):
# prints the driver's own `timing:` line and the Mcell-updates/s it amounts to.  configs[4] (32768^2, 8 tiles) and the
# multi-GPU forms of configs[3] need as many GPUs as ranks; their single-GPU parity is tests/test_gpu_virtual8.py.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
D=climate-sim-mpi-cpp_amd/driver/climate_sim_hip
for spec in "c2_4096_diffusion_periodic:4096:4096" "c3_8192_dirichlet:8192:8192" "c4_16384_2x2:16384:16384"; do
  cfg=${spec%%:*}; rest=${spec#*:}; nx=${rest%%:*}; ny=${rest##*:}
  for rep in 1 2 3; do
    out=$($D --config configs/$cfg.yaml --no-output --device-ic 2>&1 | grep "timing:")
    steps=$(grep -o "steps: [0-9]*" configs/$cfg.yaml | head -1 | grep -o "[0-9]*")
    secs=$(echo "$out" | sed -n 's/.*total_max=\([0-9.eE+-]*\) s.*/\1/p')
    python3 -c "print('$cfg rep $rep: $out  ->', round($nx*$ny*$steps/$secs/1e6), 'Mcell-updates/s ($steps steps)')"
  done
done
