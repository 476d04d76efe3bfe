# Kernel-trace summaries for profiles/: sampler-step kernels (tools/profile_swap.py) and a lone-matrix sweep
# (tools/profile_c5.py) -> gpurun_out/prof_round/*.txt
export PYTHONPATH=$PWD
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/prof_round
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_swap /tmp/pr_c5
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/pr_swap -o t -- python3 $ROOT/tools/profile_swap.py 4096 100 > $ROOT/gpurun_out/prof_round/chain_step_wall.txt 2>/dev/null || exit 1
python3 $ROOT/tools/ab/kstats.py /tmp/pr_swap colsum skinny small_kernel reduce_shares left_factor rank_update expand leaf_walk > $ROOT/gpurun_out/prof_round/chain_step_kernels.txt
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/pr_c5 -o t -- python3 $ROOT/tools/profile_c5.py 16384 > /dev/null 2>&1 || exit 1
python3 $ROOT/tools/ab/kstats.py /tmp/pr_c5 bark > $ROOT/gpurun_out/prof_round/single_matrix_n16384_kernels.txt
