// tables.h -- constant tables built on the host, see tables.cpp
#pragma once
#include <vector>
#include <cstddef>

namespace bcfgpu {
void build_errmod_tables(double depcorr, std::vector<double> &fk, std::vector<double> &beta, std::vector<double> &lhet);
void build_pl2p(double *pl2p /* [256] */);
void build_mw_table(double *mw /* [6][6][50] */);
}
