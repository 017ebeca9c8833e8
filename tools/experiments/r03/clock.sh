# GPU box: clock / power while the serial-mode bench and the VALU microbench run.  usage: bash tools/experiments/r03/clock.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R


python3 tools/clock_power.py $O/clock_serial.json -- python3 bench.py --profile-serial --steps 12 --warmup 2 > $O/serial.out 2> $O/serial.err
tail -1 $O/serial.out | cut -c1-1200
python3 tools/clock_power.py $O/clock_default.json -- python3 bench.py --steps 12 --warmup 2 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 > $O/default.out 2> $O/default.err
tail -1 $O/default.out | cut -c1-1200
