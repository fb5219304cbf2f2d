#!/bin/bash
# round 4: the write-through exchange alone, every word checked (tools/ubench/exchange_litmus.hip)
set -u
mkdir -p gpurun_out
cd tools/ubench
{
  timeout 200 ./exchange_litmus 1500000 2 0 0
  timeout 200 ./exchange_litmus 1500000 2 1 0
  timeout 200 ./exchange_litmus 600000 2 0 6
  timeout 200 ./exchange_litmus 600000 2 1 6
} 2>&1 | tee ../../gpurun_out/exchange_litmus.txt
