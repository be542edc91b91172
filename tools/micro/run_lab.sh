#!/bin/bash
# runs every conv_lab_* binary in this directory, one after the other
cd "$(dirname "$0")"
for b in conv_lab_*; do
  [ -x "$b" ] || continue
  echo "== $b"
  if [ "$b" = conv_lab_S ]; then
    for c in 0 3 7; do timeout -k 5 60 ./$b $c || exit 1; done
  else
    timeout -k 5 120 ./$b "$@" || exit 1
  fi
done
