set -e
bash tools/kernel_trace.sh c5tl product 16384 1 2 > /dev/null
head -1 gpurun_out/trace_c5tl.txt
python3 tools/ab/ksteps_period.py /tmp/kt_c5tl 3
