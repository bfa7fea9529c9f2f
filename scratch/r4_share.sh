#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4buf; mkdir -p $OUT; : > $OUT/share.log
python scratch/kbench.py "texture=" "texture_share=BODGE_AMD_STREAMED_SHARE=1" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-190 >> $OUT/share.log
python scratch/kbench.py "texture16=" "texture16_share=BODGE_AMD_STREAMED_SHARE=1" --model texture --kind z4 --vectors 16 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-190 >> $OUT/share.log
python scratch/kbench.py "potential=" "potential_share=BODGE_AMD_STREAMED_SHARE=1" --model potential --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^potential" | cut -c1-190 >> $OUT/share.log
python scratch/kbench.py "potential16=" "potential16_share=BODGE_AMD_STREAMED_SHARE=1" --model potential --vectors 16 --steps 63 --rounds 4 2>&1 | grep "^potential" | cut -c1-190 >> $OUT/share.log
python scratch/kbench.py "landau=" "landau_share=BODGE_AMD_STREAMED_SHARE=1" --model landau --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^landau" | cut -c1-190 >> $OUT/share.log
python scratch/kbench.py "ssd16=" "ssd16_share=BODGE_AMD_STREAMED_SHARE=1" --model ssd --vectors 16 --steps 63 --rounds 4 2>&1 | grep "^ssd" | cut -c1-190 >> $OUT/share.log
cat $OUT/share.log
