#!/bin/bash
# ThreadSanitizer / AddressSanitizer run of the HOST half of libdrx (csrc/drx_host.cpp: C++ PointSampler / ListSampler, the MT19937
# corruption stream, the two draw-ahead worker threads with their job rings) — CPU only, never on a GPU box.
#   bash scripts/sanitize_host.sh thread|address      -> profiles/r06_sanitize_<kind>.log
set -u
KIND=${1:-thread}
cd "$(dirname "$0")/.."
SO=$(python -m drecpy_amd.build --sanitize=$KIND | tail -1)
RT=$(gcc -print-file-name=lib$([ $KIND = thread ] && echo tsan || echo asan).so)
LOG=profiles/r06_sanitize_$KIND.log
export DRX_HOST_SANITIZER_LIB=$SO
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66"
export ASAN_OPTIONS="detect_leaks=0 exitcode=66"     # (CPython itself 'leaks' by design; leaks of the library show up as reports of its frames)
{
  echo "# $(date -u +%FT%TZ)  g++ -fsanitize=$KIND drx_host.cpp -> $SO ; LD_PRELOAD=$RT"
  echo "# python -m pytest tests/test_sampler.py -q -k 'not exports_every and not c_program and not forked_child'"
  # -R: no address-space randomisation (TSan's shadow mapping needs the classic layout on recent kernels)
  LD_PRELOAD=$RT setarch x86_64 -R python -m pytest tests/test_sampler.py -q -s -p no:cacheprovider -k 'not exports_every and not c_program and not forked_child' 2>&1
  echo "# exit code: $?"
} > $LOG 2>&1
N=$(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' $LOG)
echo "# sanitizer reports: $N" >> $LOG
tail -8 $LOG
