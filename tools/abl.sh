F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in "-DFB_ADAM_GRID=512" "-DFB_ADAM_GRID=1024" "-DFB_ADAM_GRID=2048" "-DFB_ADAM_GRID=256"; do
  (cd dqnflappybird_amd/csrc && touch fb_qnet.hip && make FLAGS="$F $v" > /dev/null 2>&1) || exit 1
  echo "== $v"; python tools/time_train.py 2>/dev/null || exit 1
done
