#!/bin/bash
# Dev: weight-gradient launches under split-K / ring-depth knobs (static getenv switches: one process per setting)
SHAPES="wgrad:128:64:256:64:1:1 wgrad:128:64:64:256:1:1 wgrad:128:64:64:64:3:1 wgrad:128:32:128:512:1:1 wgrad:128:32:128:128:3:1 wgrad:128:16:256:1024:1:1 wgrad:128:16:1024:256:1:1 wgrad:128:16:256:256:3:1 wgrad:128:8:512:512:3:1 wgrad:128:8:2048:512:1:1"
run() { echo "== $1"; env $1 python tools/time_conv.py $SHAPES 2>&1 | grep median; }
run "X=0"
run "SAT_SPLIT_TARGET128=1024 SAT_SPLIT_TARGET=1536"
run "SAT_SPLIT_TARGET128=256 SAT_SPLIT_TARGET=384"
run "SAT_GLDS_STAGES=3"
run "SAT_GLDS_STAGES=4"
run "SAT_GLDS_STAGES=3 SAT_SPLIT_TARGET128=1024 SAT_SPLIT_TARGET=1536"
run "SAT_WIDE_TILES=0"
run "SAT_SPLIT_FRAC=2"
run "SAT_SPLIT_FRAC=32"
