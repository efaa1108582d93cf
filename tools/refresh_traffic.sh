#!/bin/bash
# profiles/traffic_latest.json for the sources as they are (its source_sha256 must match tools/source_hash.py, or bench.py reports
# traffic: null): the PMC passes of the c3 probe-side pass only.  usage (GPU box): bash tools/refresh_traffic.sh  ->  gpurun_out/traffic_latest.json
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out/tr
./tools/pmc.sh gpurun_out/tr/pmc_c3 --workload c3 --no-host-entry > gpurun_out/tr/c3_pmc_summary.txt 2>&1
python tools/traffic_json.py gpurun_out/tr/c3_pmc_summary.txt gpurun_out/traffic_latest.json
cat gpurun_out/traffic_latest.json; python tools/source_hash.py | tail -1
