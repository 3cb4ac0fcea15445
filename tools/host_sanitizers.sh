#!/bin/bash
# Development aid: the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (device code is
# compiled as usual: sanitizers for GPU code are not available on this pool). Two programs use it: the C++
# driver (per-particle maps, the whole front-end; synthetic frames and the replay of the reference's run) and
# tools/san_shared_loop.cpp (random shared-grid steps, 1 to 70 000 particles).
#   bash tools/host_sanitizers.sh build     here, on the CPU: everything into tools/_san/ (not committed)
#   bash tools/host_sanitizers.sh run       on the GPU box (gpurun): prints the number of sanitizer reports
# The HSA runtime's own teardown trips an ASan CHECK after main() has returned (sanitizer_allocator_device.h);
# that is the runtime unloading, not this library: the count below looks at reports only.
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
    mkdir -p tools/_san
    SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
    (cd slam_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $SAN -shared \
        -o ../../tools/_san/libslam_pf.so pf_kernels.hip pf_resample.hip pf_api.hip pf_static.hip pf_measure_box.hip pf_fuse_cols.hip pf_host.cpp) || exit 1
    for prog in driver/step_loop.cpp tools/san_shared_loop.cpp; do
        /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer \
            -o tools/_san/$(basename $prog .cpp) $prog -Ltools/_san -lslam_pf -Wl,-rpath,'$ORIGIN' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 1
    done
    ls -la tools/_san
    exit 0
fi
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
mkdir -p gpurun_out
python3 tools/mk_replay.py /tmp/frames.replay || exit 1
timeout -k 10 200 ./tools/_san/step_loop replay /tmp/frames.replay > gpurun_out/san_replay.out 2> gpurun_out/san_replay.err
timeout -k 10 300 ./tools/_san/step_loop 6 5000 > gpurun_out/san_synth.out 2> gpurun_out/san_synth.err
timeout -k 10 300 ./tools/_san/san_shared_loop > gpurun_out/san_shared.out 2> gpurun_out/san_shared.err
for f in replay synth shared; do
    echo "$f: $(grep -c 'runtime error\|ERROR: AddressSanitizer' gpurun_out/san_$f.err) sanitizer reports; last line: $(tail -n 1 gpurun_out/san_$f.out)"
done
