#!/bin/bash
# round 4: the write-through exchange alone, every word checked (tools/ubench/exchange_litmus.hip; built in the container:
#   hipcc --offload-arch=gfx950 -O3 -o exchange_litmus exchange_litmus.hip ; ... -DLITMUS_S=12 -o exchange_litmus_s12 ...)
set -u
mkdir -p gpurun_out
cd tools/ubench
{
  timeout 300 ./exchange_litmus_s12 2500000 2 0 0 20
  timeout 300 ./exchange_litmus_s12 1200000 2 0 6 20
  timeout 300 ./exchange_litmus 2000000 2 0 0
} 2>&1 | tee ../../gpurun_out/exchange_litmus2.txt
