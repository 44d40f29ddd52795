run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 10 2>&1 | grep -E "\"value\"|Error|error" | sed -E 's/.*"value": ([0-9.]+).*"ms_per_step": ([0-9.]+).*"render_mpix_per_s": ([0-9.]+).*"flashsplat_views_per_s": ([0-9.]+).*"final_loss": ([0-9.]+).*/it\/s \1 ms \2 mpix \3 flash \4 loss \5/'; }
run A=1
run W3D_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
