cd $GRAFT_REPO_ROOT
timeout 600 bash tools/profile.sh r06_final > gpurun_out/r06_final_profile.log 2>&1
N=16000000 timeout 600 bash tools/profile_inflate_pmc.sh r06_final > gpurun_out/r06_final_inflate_pmc.log 2>&1
rm -f /tmp/one.fq.gz /tmp/one.fq; N=2000000 GZL=6 timeout 400 bash tools/gunzip_profile.sh r06_final > gpurun_out/r06_final_gunzip_profile.log 2>&1
timeout 300 bash tools/gz_modes_check.sh > gpurun_out/r06_final_gz_modes.txt 2>&1
N=8000000 SINK=/dev/null ENVS="X=1" timeout 300 bash tools/packed_e2e.sh r06_final_null > /dev/null 2>&1
N=8000000 ENVS="X=1" timeout 300 bash tools/packed_e2e.sh r06_final_file > /dev/null 2>&1
N=8000000 N1=2000000 GZL=6 QUICK=1 timeout 600 bash tools/gz_e2e.sh r06_final > /dev/null 2>&1
timeout 500 bash tools/exit_probe.sh > gpurun_out/r06_final_exit_probe.txt 2>&1
timeout 900 bash tools/c3_e2e.sh > gpurun_out/r06_final_c3_e2e.txt 2>&1
ls -la gpurun_out | tail -30
