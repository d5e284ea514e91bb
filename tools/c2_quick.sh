for n in 64 256 2048; do python tests/soak.py 60000 demc $n 2>&1 | grep -o "nchains.*\|([0-9.]* M/s" | tr '\n' ' '; echo; done
