#!/bin/bash
# run every example script once (drop-in API on the device)
for f in pendulum.py scissor.py pend_on_cart_optimization.py puppet_optimization.py puppet_basic.py dual_pendulums.py extensor_tendon.py batch_discopt.py; do
  echo "== $f"; timeout 600 python examples/$f 2>&1 | tail -4
done
