#!/bin/bash
# everything the round's evidence consists of, in one call: GPU suite, smoke, then scratch/r2_profiles.sh
set -o pipefail
bash scratch/final_tests.sh || exit 1
bash scratch/r2_profiles.sh
