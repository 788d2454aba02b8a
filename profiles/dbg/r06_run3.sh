# round 6: variants' launch times + bit identity of the product + stamps of the diag build
cd $GRAFT_REPO_ROOT
TAG=${1:-r06d}
FORMS=one,duo,duo_pairs bash profiles/dbg/r06_variants.sh ${TAG}_variants "$2" "product"
RG_DIAG=1 timeout 300 python profiles/dbg/seq2_stamps.py 64 0 > gpurun_out/${TAG}_seq2_stamps.txt 2>&1
head -4 gpurun_out/${TAG}_seq2_stamps.txt | cut -c1-420
sed -n 5,26p gpurun_out/${TAG}_seq2_stamps.txt
