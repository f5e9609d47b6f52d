# Per-wave records of ONE k_step_roam launch (lifetime, live lanes, pairs, slow pairs, strays, window):
# builds tracks.hip with -DSSRS_DEBUG_WAVE_DUMP=<launch> into ssrs_amd/libssrs_dump_tmp.so (HERE, hipcc), then on the
# GPU box:  SSRS_HIP_LIB=$GRAFT_REPO_ROOT/ssrs_amd/libssrs_dump_tmp.so SSRS_TRACKS_DEBUG_ROAM=1 python bench.py --steps 1 \
#           --warmup 0 --cpu-seconds 0 --stand-in-steps 0 --no-chain-probe 2> gpurun_out/dump.err
# and tools/dev/r03_wave_dump.py gpurun_out/dump.err prints the analysis (profiles/r03_roam_waves.txt).
set -e
cd "$(dirname "$0")/../../ssrs_amd/csrc"
LAUNCH=${1:-52}
python build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -DSSRS_DEBUG_WAVE_DUMP=$LAUNCH -c tracks.hip -o /tmp/tracks_dump.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libssrs_dump_tmp.so capi.o raster.o /tmp/tracks_dump.o presence.o potential.o thermals.o amg.o -ldl
echo ../libssrs_dump_tmp.so
