for v in "" 1; do for s in 1 2 3; do NS_NGP_GRID_DECAY_ALL=$v python tools/ngp_bench.py 100 700 2>/dev/null | tail -1; done; echo "--- decay_all=$v"; done
