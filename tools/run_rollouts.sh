#!/bin/bash
# GPU box: multi-step rollout configs at local batch 2: ms/step, samples/s, peak memory, for the in-place rollout +
# selective checkpointing against the reference-shaped torch.cat rollout + torch.utils.checkpoint
cd $GRAFT_REPO_ROOT
O=gpurun_out/rollouts.txt; : > $O
for cfg in bench_depth12_e128_2step bench_depth12_e128_4step bench_depth12_e128_8step; do
  SWV2_ROLLOUT_INPLACE=1 python tools/run_cfg.py $cfg 2 4 2>&1 | tail -1 | sed 's/^/inplace + selective ckpt (fp32 inputs): /' >> $O
  SWV2_ROLLOUT_INPLACE=1 SWV2_CKPT_BF16=1 python tools/run_cfg.py $cfg 2 4 2>&1 | tail -1 | sed 's/^/inplace + selective ckpt (bf16 inputs): /' >> $O
  SWV2_ROLLOUT_INPLACE=0 SWV2_CKPT_TORCH=1 python tools/run_cfg.py $cfg 2 4 2>&1 | tail -1 | sed 's/^/torch.cat rollout + torch.utils.checkpoint:  /' >> $O
done
cat $O
