// TEST INFRASTRUCTURE (oracle/): force-included in front of the reference's own text when
// oracle/build_ref.sh compiles the ROS-free line ranges of /root/reference/Slam_3d_particle.cpp.
//
// It adds nothing the image lacks: <random>, <cstring> and <vector> are the system's own headers (the
// reference calls memcpy and uses std::vector without including them once its ROS headers are left out). The only thing it pins is the seed
// source: the reference seeds every mt19937 from std::random_device (S3P:768, S3P:852), which
// would make its outputs irreproducible. The macro below makes the unqualified name
// `random_device` in the reference's text resolve to a functor that hands out seeds the
// harness chose, so the harness knows the seed of every engine the reference constructs.
#pragma once
#include <random>
#include <cstring>
#include <vector>

extern "C" unsigned pf_ref_next_seed;          // defined in ref_harness_tail.cpp
struct pf_fixed_random_device {
    unsigned operator()() { return pf_ref_next_seed++; }
};
#define random_device pf_fixed_random_device
