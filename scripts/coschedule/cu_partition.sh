#!/bin/bash
# $1 = frame-loop CUs (low bits), rest to the vocoder; $2 = layout (lo|alt)
X=$1
if [ "$2" = "alt" ]; then
  # interleaved: frame loop takes bit i where (i % 256*... ) pattern by python
  MA=$(python3 -c "X=$X; m=0
acc=0
for i in range(256):
    acc+=X
    if acc>=256: acc-=256; m|=1<<i
print(hex(m))")
else
  MA=$(python3 -c "print(hex((1<<$X)-1))")
fi
MB=$(python3 -c "print(hex(((1<<256)-1) ^ $MA))")
rm -f /tmp/voc_ready /tmp/frame_ready
ROC_GLOBAL_CU_MASK=$MB python scripts/coschedule/part_voc.py 8 $3 2>&1 | grep VOC &
ROC_GLOBAL_CU_MASK=$MA python scripts/coschedule/part_frame.py 4 2>&1 | grep FRAME
wait
