#!/bin/bash
# drop-in, one sequence at a time with PNGs: writers per directory (EVREAL_PNG_PER_DIR), three calls each on one box
for rep in 1 2 3; do for pd in 4 6 8 12; do
EVREAL_PNG_PER_DIR=$pd python bench.py --config eval_cli --sub 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('per_dir=$pd', 'on', d['save_images_on']['value'], 'seq1_on', d['one_sequence_at_a_time_save_images_on']['value'], 'seq1', d['one_sequence_at_a_time']['value'])"
done; done
