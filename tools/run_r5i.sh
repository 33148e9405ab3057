# round 5: (1) counter calibration; (2) kernel trace of a ResNet-50 step (per-grid durations of the pointwise input gradient with the addend); (3) full suite
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/calib $R/gpurun_out/prof_r5i_r50
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib/fetch -o run -- $R/tools/csrc/fetch_calib > $R/gpurun_out/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib/write -o run -- $R/tools/csrc/fetch_calib > $R/gpurun_out/calib_write.log 2>&1
python $R/tools/fetch_calib_table.py $R/gpurun_out/calib > $R/gpurun_out/r05_fetch_calib.md; cat $R/gpurun_out/r05_fetch_calib.md
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5i_r50 -o run -- python $R/tools/net_bench.py --arch resnet50 --steps 3 > $R/gpurun_out/prof_r5i_r50.log 2>&1
python $R/tools/kernel_by_grid.py $R/gpurun_out/prof_r5i_r50/run_results.db 'k_pw<.*, true, false, (true|false)>' > $R/gpurun_out/r5i_pw_by_grid.md; cat $R/gpurun_out/r5i_pw_by_grid.md
rm -rf $R/gpurun_out/calib/*/*/*.db $R/gpurun_out/prof_r5i_r50
cd $R
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_r5i.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_r5i.log | tail -3
