python tools/voxel_bench.py --windows 64 512 2048 2>/dev/null | tail -4
for st in 20 100; do python bench.py --sub --steps $st --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); rv = d['roofline_voxelizer']; print('steps $st |', d['value'], rv['in_step'], rv.get('standalone_512'))"; done
