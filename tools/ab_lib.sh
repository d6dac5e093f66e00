#!/bin/bash
# A/B of two builds of the library on one box: porla_amd/_ab/libA.so and libB.so are copied over porla_amd/libmultiexp.so in turn
# usage: tools/ab_lib.sh '<bench command>' [reps]
CMD="$1"; REPS=${2:-2}
for rep in $(seq $REPS); do
  for v in A B; do
    cp porla_amd/_ab/lib$v.so porla_amd/libmultiexp.so
    echo "== $v (rep $rep)"
    bash -c "$CMD"
  done
done
cp porla_amd/_ab/libA.so porla_amd/libmultiexp.so
