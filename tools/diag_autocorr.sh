# kernel timing experiment (DIAG build only): isolated duration of srla_autocorr cut short at successive points
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag_ac
mkdir -p $O
rm -f $O/result.txt
for v in 0 1; do for st in 11 12 13 14 15 16 0; do
  export SRLA_MI355X_FUSED_FFT=$v
  if [ $st = 0 ]; then unset SRLA_MI355X_K3_STOP; else export SRLA_MI355X_K3_STOP=$st; fi
  rm -rf /tmp/dd; timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d /tmp/dd -o run -- python $R/tools/perf_probe.py 174.8 device 1 ${1:-1} ${2:-0} ${3:-4096} > /tmp/dd.log 2>&1
  echo "variant $v stop $st" >> $O/result.txt
  python $R/tools/summarize_pmc.py /tmp/dd | python -c "import csv,sys; [print(r[0][:34], r[1], r[2]) for r in csv.reader(sys.stdin) if 'autocorr' in r[0]]" >> $O/result.txt
done; done
cat $O/result.txt
