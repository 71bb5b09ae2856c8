#!/bin/bash
# tools: libbsx_nofast.so = the product library with -DBSX_NO_SYNC_FAST_PATH (every synchronous bsx_header_range on a coalescing context goes
# through the batcher; the lone-caller serial path off) for the A/B in tools/lone_caller_ab.py.  Never shipped.
set -e
cd "$(dirname "$0")/../blobstreamx_amd/csrc"
mkdir -p build_nofast
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -DBSX_NO_SYNC_FAST_PATH -c api.hip -o build_nofast/api.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libbsx_nofast.so $(ls build/*.o | grep -v "/api.o") build_nofast/api.o
ls -la ../lib/libbsx_nofast.so
