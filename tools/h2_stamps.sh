#!/bin/bash
# phase split of the two-role head from the stamped variant (tools/build_variant.sh h2stamps "-DNIS_WM=32 -DH2_STAMPS" flow_bwd_wm flow_bwd_wm32)
NIS_LIB=$PWD/nf_amd/lib/variants/h2stamps.so python3 tools/train_probe.py C5 2e5 1 2 batched 2>&1 | grep -E "^h2|^C5" | tail -8
