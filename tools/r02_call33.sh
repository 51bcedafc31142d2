#!/bin/bash
# Where does the feed's time go on a 3 Gbp FASTA in /dev/shm?  JFGPU_FEED_TRACE with different numbers of pread streams.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
  oracle/_ref/ref_generate_sequence -s 42 -r 150 -o /dev/shm/feed3g 3000000000
  ls -l /dev/shm/feed3g.fa
  export JFGPU_QUIET=1 JFGPU_FEED_TRACE=1
  for reg in 1 0 1; do
    echo "== JFGPU_FEED_REGISTER=$reg"
    JFGPU_FEED_REGISTER=$reg bin/jellyfish-amd count -m 21 -C -s 5G --no-write --digest /dev/stdout --timing /dev/stdout /dev/shm/feed3g.fa 2>&1 | grep -v "amdgpu.ids"
  done
  echo "== register, chunk 128 MiB"
  JFGPU_PARSE_CHUNK=134217728 bin/jellyfish-amd count -m 21 -C -s 5G --no-write --timing /dev/stdout /dev/shm/feed3g.fa 2>&1 | grep -v "amdgpu.ids"
  echo "== a file on disk (not tmpfs)"
  cp /dev/shm/feed3g.fa /tmp/feed3g.fa && JFGPU_FEED_REGISTER=1 bin/jellyfish-amd count -m 21 -C -s 5G --no-write --timing /dev/stdout /tmp/feed3g.fa 2>&1 | grep -v "amdgpu.ids"; rm -f /tmp/feed3g.fa
  echo "== taskset: what CPUs may this shell use?"; taskset -p $$; cat /sys/fs/cgroup/cpu.max 2>/dev/null
  rm -f /dev/shm/feed3g.fa
} > gpurun_out/r02_call33.log 2>&1
cat gpurun_out/r02_call33.log | cut -c1-400
