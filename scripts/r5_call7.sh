# round 5, seventh GPU call: the GPU suite after the solver-side prune and the batched guard; 100-concept soaks under host switches
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r05_t7.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t7.txt
tail -4 gpurun_out/r05_t7.txt
{
python scripts/soak_n.py 100 400
EMCID_EARLY_VSTAR=0 python scripts/soak_n.py 100 400
python scripts/soak_n.py 100 400 --gc off
python scripts/soak_n.py 100 400 --gc freeze
EMCID_TOK_THREADS=1 python scripts/soak_n.py 100 400
EMCID_WEIGHT_GUARD=0 python scripts/soak_n.py 100 400
python scripts/soak_n.py 1000 300
python scripts/soak_n.py 1000 300 --gc freeze
EMCID_WEIGHT_GUARD=0 python scripts/soak_n.py 1000 300
} > gpurun_out/r05_n100_soak.txt 2>&1
cat gpurun_out/r05_n100_soak.txt | grep "^N "
echo done
