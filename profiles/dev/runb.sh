timeout 600 python -m pytest tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -3
WC_MATCH_DEBUG=1 WC_MATCH_TIMING=1 timeout 1200 python bench.py --steps 20 --no-cpu-baseline --no-clouds > gpurun_out/b1.json 2> gpurun_out/b1.err
grep "tree depth" gpurun_out/b1.err | sort -u -k3,8 | head -8; grep "k_knn_tree by" gpurun_out/b1.err | tail -4
python - <<EOP
import json
r=json.load(open("gpurun_out/b1.json"))
w=r["window"]; o=r["odometry_step"]
print("window match_s",w["match_s"],"M surf/s",w["match_surfels_per_s"]/1e6,"room",w["match_room_stream"]["ms_per_search"],w["match_room_stream"]["ms_per_50k_queries"])
print("step",o["ms_per_step"],o["stage_ms"])
EOP
