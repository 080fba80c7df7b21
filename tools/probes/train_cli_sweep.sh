#!/bin/bash
# Functional sweep of gcn_vae_amd.train over flag combinations on a small synthetic data set: every run must finish with a finite
# loss and an MRR line (catches crashes in corners the unit tests do not combine).  Not a timing.
cd "$(dirname "$0")/../.."
D="synthetic:600:11:5000:200:200:2"
run() { tag=$1; shift; timeout -k 10 240 python -m gcn_vae_amd.train -d ${DS:-$D} --gpu 0 --n-hidden 16 --n-bases 4 --n-epochs 6 --evaluate-every 3 \
          --graph-batch-size 800 --eval-batch-size 50 --model-state-file /tmp/ms_$tag.pth "$@" > /tmp/tr_$tag.log 2>&1; rc=$?
        echo "$tag rc=$rc $(grep -c -i 'nan' /tmp/tr_$tag.log) nan-lines; $(grep -i 'mrr' /tmp/tr_$tag.log | tail -1 | cut -c1-80)"; [ $rc -ne 0 ] && tail -5 /tmp/tr_$tag.log; }
run plain
run flows2 --n-flows 2 --mmd-param 1.0
run flows2bf16 --n-flows 2 --mmd-param 1.0 --bf16
run nbr --edge-sampler neighbor
run dev --device-sampler
run devgraph --device-sampler --graph-step --n-flows 1 --mmd-param 1.0
run devnbr --device-sampler --edge-sampler neighbor --graph-step
run rgcn --model-class RGCN
DS="synthetic:600:50:5000:200:200:2" run h200 --n-hidden 200 --n-bases 100 --n-flows 3 --mmd-param 1 --kl-param 1e-3
run load --load True --model-state-file /tmp/ms_plain.pth
run testmode --test-mode True --model-state-file /tmp/ms_plain.pth
