run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 8 --all-stages 2>&1 | grep -E "tile_count|fill_lists|\"value\"" | sed -E 's/.*"value": ([0-9.]+).*"ms_per_step": ([0-9.]+).*/it\/s \1 ms \2/'; }
run A=1
run W3D_TUNE_BAND_TILES_FILL=1024
run W3D_TUNE_BAND_TILES=2048
