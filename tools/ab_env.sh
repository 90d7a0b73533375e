#!/bin/bash
# A/B of an environment switch on the driver-shaped batch, alternating on ONE box:   tools/ab_env.sh VAR=value [rounds]
set -u
SW=$1; R=${2:-3}
for i in $(seq $R); do
  python3 tools/ab_kernel.py mixed 6 | sed "s/^/default      /"
  env $SW python3 tools/ab_kernel.py mixed 6 | sed "s/^/$SW /"
done
