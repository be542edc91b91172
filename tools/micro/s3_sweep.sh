#!/bin/bash
# persistent 3x3 kernel (conv_s3.h) instantiations on the 64-channel layers: LAB_S3 forces one, the column
# "[0,0,0,0,0,0]" of conv_lab's output is its time (the first number is the generic kernel's)
cd "$(dirname "$0")"
for ld in 1 4; do
for cfg in "4,2,1,4,256,2" "4,2,1,4,192,3" "4,2,1,4,192,2" "4,1,1,8,384,2" "4,1,1,4,192,3" "4,1,1,4,256,2"; do
  echo "== LAB_S3=$cfg ldmul=$ld"
  for c in 8 16 29; do LAB_QUICK=1 LAB_S3=$cfg timeout -k 5 60 ./conv_lab_A $c $ld | grep -v "^total" || exit 1; done
done
done
