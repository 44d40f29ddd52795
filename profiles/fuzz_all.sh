# The differential fuzzers of profiles/fuzz_*_probe.py from fresh seeds in one gpurun call -> gpurun_out/fuzz_r06.txt.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fuzz_r06.txt
: > $OUT
run() { echo "== $*" >> $OUT; ( time timeout $1 python3 profiles/$2 $3 $4 ) >> $OUT 2>&1; }
run 400 fuzz_share_probe.py 6000 200000
run 300 fuzz_culls_probe.py 4000 200000
run 400 fuzz_oracle_probe.py 4000 200000
run 300 fuzz_grads_probe.py 1000 200000
run 200 fuzz_flash_probe.py 1000 200000
run 200 fuzz_densify_probe.py 1000 200000
run 300 fuzz_step_probe.py 200 200000
grep -v "amdgpu.ids\|^$\|user\|sys" $OUT
