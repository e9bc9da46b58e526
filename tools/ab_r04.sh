#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_TURN_MS=1000 python3 tools/ab_variants.py mixed-6x64 65536 5 main nox xphaseA main nox xphaseA main nox xphaseA 2>/dev/null
AB_TURN_MS=1000 python3 tools/ab_variants.py dense-6x64 65536 5 main nox xphaseA main nox xphaseA main nox xphaseA 2>/dev/null
AB_TURN_MS=1000 python3 tools/ab_variants.py stress-12x128 16384 5 main nox main nox main nox 2>/dev/null
