for Q in 8 12 24; do
  GPU_MAX_HW_QUEUES=$Q KW="{}" TAG=q$Q KS="20" bash profiles/dbg/sweep_kwargs.sh
done
KW="{}" TAG=q16 KS="20" bash profiles/dbg/sweep_kwargs.sh
