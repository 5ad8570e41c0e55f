set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
VSV_BAM_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/p_ing -- python tools/ingest_bench.py 3000000 0 --device-only > gpurun_out/r2f/ingest_3m.log 2>&1
rm -f gpurun_out/r2f/p_ing/*/*.db
cp gpurun_out/r2f/p_ing/*/*kernel_stats.csv gpurun_out/r2f/r02_ingest_device_3m_kernel_stats.csv
rm -rf gpurun_out/r2f/p_ing
grep -v "^W2026\|rocprofv3\|amdgpu.ids" gpurun_out/r2f/ingest_3m.log | tail -12
