// pcg_kernels.hpp -- placeholder until the PCG kernels land (see pcg_abi.inc).
#pragma once
