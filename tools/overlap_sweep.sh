#!/bin/bash
# How much do the extractor's internal frame-range streams (--variant streams=N), the side-stream blur (--variant side_blur=N) and the
# runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4) change the step time?  One bench run per setting.
run() { python3 bench.py --no-cpu-baseline --no-density-sweep --no-extra-configs --steps 20 $1 | python3 -c "import json,sys; d=json.load(sys.stdin); print('%9.0f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for Q in 4 8 16; do for S in 1 2 4; do
  export GPU_MAX_HW_QUEUES=$Q; echo -n "hw queues $Q, streams $S, side blur 1: "; run "--variant streams=$S --variant side_blur=1"
done; done
export GPU_MAX_HW_QUEUES=4; echo -n "hw queues 4, streams 1, side blur 0: "; run "--variant side_blur=0"
export GPU_MAX_HW_QUEUES=8; echo -n "hw queues 8, streams 1, batch 512: "; run "--batch 512"
