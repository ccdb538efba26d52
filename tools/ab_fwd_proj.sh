#!/bin/bash
# developer tool (GPU box): tools/time_fwd_proj.py with every library under variants/ (timing-only ablation builds of the
# projection epilogue: -DFWD_ABL=<bits>, see csrc/mh_lbs.hip)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  python tools/time_fwd_proj.py 2>/dev/null | tail -1
  for f in variants/lib*.so; do MHHIP_LIB=$GRAFT_REPO_ROOT/$f python tools/time_fwd_proj.py 2>/dev/null | tail -1; done
done
