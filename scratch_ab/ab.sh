#!/bin/bash
for v in old new old new; do
  cp scratch_ab/libvit_hash_$v.so vit_amd/libvit_hash.so || exit 1
  echo "== $v"
  timeout -k 10 120 python -u tools/probe_deflate.py 32768 4096 300 | grep "^{" | cut -c60-130 || exit 1
  timeout -k 10 120 python -u tools/probe_deflate.py 1000000 4096 300 | grep "^{" | cut -c60-130 || exit 1
  timeout -k 10 300 python -u tools/probe_deflate_text.py 32768 4096 | cut -c85-200 || exit 1
  timeout -k 10 300 python -u tools/probe_deflate_text.py 262144 4096 | cut -c85-200 || exit 1
done
cp scratch_ab/libvit_hash_new.so vit_amd/libvit_hash.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_deflate.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
