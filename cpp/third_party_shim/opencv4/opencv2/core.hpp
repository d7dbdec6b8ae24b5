// empty stand-in: reference test/test_ba.cpp:9-11 includes the OpenCV headers
// but uses no cv:: symbol; only used when OpenCV is not installed.
#pragma once
