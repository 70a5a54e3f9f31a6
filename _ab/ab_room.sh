for r in 1 2; do for v in Old New; do cp _ab/lib$v.so wildcat-slam_amd/csrc/libwildcat_hip.so; python bench.py --no-clouds --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['window']; print('$v', w['match_surfels_per_s'], w['match_room_stream']['ms_per_search'])"; python profiles/dev/step_var.py 16 | head -1 | cut -c1-60; done; done
