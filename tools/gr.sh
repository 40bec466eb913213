#!/bin/bash
# Build, check that the library exports what include/bt_hip.h declares, then hand the command to gpurun (never ship a stale .so).
set -e
cd "$(dirname "$0")/.."
make -C bayesian_torch_amd/csrc -j8 > /tmp/bt_make.log 2>&1 || { tail -30 /tmp/bt_make.log; exit 1; }
make -C oracle > /dev/null 2>&1 || true
python -m pytest tests/test_api_surface.py -x -q -k "exports or struct" > /tmp/bt_api.log 2>&1 || { tail -20 /tmp/bt_api.log; exit 1; }
to=${GR_TIMEOUT:-900}
exec /usr/local/graft/bin/gpurun --timeout $to -- "$@"
