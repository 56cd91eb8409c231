#!/bin/bash
# window budgets re-swept with COLD pools
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3al; mkdir -p $O
for rep in 1 2; do
  timeout -k 10 400 python experiments/ab_case.py - c1@4224 c1@5632 c1@7168 c1@9216 c1@12288 c3@4224 c3@5632 c3@7168 c3@9216 c3@12288 c1:8@5632 c1:8@7168 c1:8@12288 c3:8@5632 c3:8@7168 c3:8@12288 2>> $O/ab.err | cut -c24-160 >> $O/ab.log
done
cat $O/ab.log
