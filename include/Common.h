// Common.h — shared libc / STL includes for the host mirror of the Quantized-CNN interface.
// Same role as the reference's include/Common.h:9-26; nothing here is device code.
#ifndef QCNN_HOST_COMMON_H_
#define QCNN_HOST_COMMON_H_

#include <assert.h>
#include <float.h>
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <iostream>
#include <string>
#include <typeinfo>
#include <vector>

#endif  // QCNN_HOST_COMMON_H_
