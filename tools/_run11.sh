for n in 7 10 14 16 21 28 32 48; do echo -n "chunks $n: "; bash tools/_q.sh --orb-chunks $n; done
