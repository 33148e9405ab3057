# round 5: the whole GPU suite, the 3-task sequence at full size (quick form first), the grown bench under rocprofv3
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_r5f.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_r5f.log | tail -5
python bench.py --task-sequence 3 --steps 22 --batch 32 > gpurun_out/r5f_seq_quick.log 2>&1; tail -1 gpurun_out/r5f_seq_quick.log | cut -c1-600
python bench.py --task-sequence 3 --steps 220 > gpurun_out/r5f_seq.log 2>&1; tail -1 gpurun_out/r5f_seq.log | cut -c1-300
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5f_grown
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5f_grown -o run -- python $R/bench.py --width-multiplier 1.5 --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/prof_r5f_grown.log 2>&1
ls $R/gpurun_out/prof_r5f_grown | head
