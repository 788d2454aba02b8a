# PMC passes of rg_seq2_kernel alone on the chip, both launch forms: bash profiles/dbg/pmc_seq2_forms.sh <prefix>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
P=$1
for FORM in ${FORMS:-wide pairs}; do
  if [ $FORM = wide ]; then export SEQ_PAIRS=0 SEQ_DUO=1; else export SEQ_PAIRS=1 SEQ_DUO=1; fi
  D=gpurun_out/pmc7_$FORM
  rm -rf $D && mkdir -p $D
  rocprofv3 --kernel-trace --stats -d $D/seq_stats --output-format csv -- python3 profiles/pmc_seq.py > $D/info.txt 2> $D/e1.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e2.txt
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/seq_write --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e3.txt
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $D/seq_mfma --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e4.txt
  python3 profiles/pmc_seq_summarize.py $D "$(grep '^{' $D/info.txt | tail -1)" gpurun_out/${P}_pmc_seq2_$FORM.json > gpurun_out/${P}_pmc_seq2_$FORM.txt 2>&1
  tail -4 gpurun_out/${P}_pmc_seq2_$FORM.txt
  find $D -name "*.csv" -size +3M -delete
done
