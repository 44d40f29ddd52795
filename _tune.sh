run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 10 --all-stages 2>&1 | grep -E "stage\]|\"value\"" | sed -E 's/.*"value": ([0-9.]+).*"ms_per_step": ([0-9.]+).*"render_mpix_per_s": ([0-9.]+).*"flashsplat_views_per_s": ([0-9.]+).*/it\/s \1 ms \2 mpix \3 flash \4/'; }
run A=1
