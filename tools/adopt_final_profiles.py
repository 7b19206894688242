"""Copy what tools/gpu_r6_final.sh left under gpurun_out/<tag>/ into profiles/ and refresh profiles/traffic.json from the same-run
PMC figures of the driver-command bench line: python tools/adopt_final_profiles.py gpurun_out/r06final5"""
import json
import os
import shutil
import sys

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = ["r06_driver_cmd_bench.json", "r06_driver_cmd_rocprof.txt", "r06_pmc_rollout.txt", "r06_pmc_rollout_config4.txt", "r06_sq_counters_config4.json",
         "r06_pytest_gpu.log", "r06_bench_config3.json", "r06_bench_config4.json", "r06_bench_config5.json", "r06_bench_1M_envs.json",
         "r06_force_dist_nccl_1rank.json", "r06_single_process_2shards_1gpu.json", "r06_single_env_final.txt",  # (r06_soak.log: the 300-seed run is kept)
         "r06_two_ranks_one_gpu_gloo.json", "sq_counters.json", "r06_soak_step_server.log"]
for n in names:
    p = os.path.join(src, n)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(root, "profiles", n))
    else:
        print("missing", n)
r = json.load(open(os.path.join(src, "r06_driver_cmd_bench.json")))
ts = r["roofline"]["traffic_source"]
t = {"_kernel_source_sha": ts["kernel_source_sha"],
     "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate child passes of `python bench.py --gpus 1 --steps 20 --warmup 5` (profiles/"
              "r06_driver_cmd_bench.json, the same-run figures copied here as the fallback for runs that cannot collect counters); KiB -> bytes; "
              "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half); WRITE_SIZE as reported",
     "k_rollout5<true, true, false, false, false>": {"FETCH_SIZE_bytes_per_launch": ts["fetch_bytes"], "WRITE_SIZE_bytes_per_launch": ts["write_bytes"],
                                                     "hbm_bytes_per_launch": ts["fetch_bytes"] + ts["write_bytes"], "launches": 3}}
json.dump(t, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
sq = json.load(open(os.path.join(src, "sq_counters.json")))
print("sha bench %s / sq %s" % (ts["kernel_source_sha"], sq.get("kernel_source_sha")))
print(json.dumps(r["summary"]))
