#!/bin/bash
# scratch driver for one gpurun call (rewritten per call; see tools/final_runs.sh for the set a round keeps)
bash tools/final_runs.sh scratch < /dev/null
