// tPatchGNN LearnableTE + TTCN patch encoder: see ttcn.hip (entry points are declared in include/immtsf.h).
#pragma once
#include "common.hpp"
