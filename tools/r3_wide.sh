#!/bin/bash
for w in 0 1; do for c in c4 c3 c2; do echo "--- $c SAT_WIDE_TILES=$w"; SAT_WIDE_TILES=$w CFG=$c N=10 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"; done; done
