// Forced-include for building the REFERENCE's own XUSG/Optional/XUSGObjLoader.cpp with g++.
// The reference relies on its MSVC precompiled header (stdafx.h) for the standard headers and on
// three Annex-K/MSVC CRT names.  Nothing is re-implemented here: the names map one-to-one onto the
// ISO C functions of the same behaviour (the extra buffer-size argument of "%s" is ignored by
// fscanf, exactly as if it were not there).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define fscanf_s fscanf
#define sscanf_s sscanf
#define fopen_s(pp, name, mode) ((*(pp) = fopen((name), (mode))) ? 0 : 1)
