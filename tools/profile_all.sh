#!/bin/bash
# All rocprofv3 passes of the three benchmark configurations (GPU box): tools/profile_all.sh <round tag, e.g. r03>
R=${1:-r03}
bash tools/profile.sh ${R}c3 && bash tools/profile_cfg.sh ${R}_c4 c4 3840x2160 && bash tools/profile_cfg.sh ${R}_c5 c5 7680x4320
