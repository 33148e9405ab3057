# FETCH_SIZE / WRITE_SIZE against known byte counts (profiles/r05_fetch_calib.md): build tools/csrc/fetch_calib.hip first
# (hipcc --offload-arch=gfx950 -O2 -o tools/csrc/fetch_calib tools/csrc/fetch_calib.hip), then one counter per pass, kernel trace only.
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/calib
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib/fetch -o run -- $R/tools/csrc/fetch_calib > $R/gpurun_out/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib/write -o run -- $R/tools/csrc/fetch_calib > $R/gpurun_out/calib_write.log 2>&1
python $R/tools/fetch_calib_table.py $R/gpurun_out/calib > $R/gpurun_out/r05_fetch_calib.md; cat $R/gpurun_out/r05_fetch_calib.md
rm -rf $R/gpurun_out/calib/*/*/*.db
