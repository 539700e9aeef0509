timeout 900 python -m pytest tests/test_refined_gpu.py -x -q -k "tile_maps or cell_sort" 2>&1 | tail -15
python3 tools/check_tilemaps.py 128 2>&1 | grep "maps="
