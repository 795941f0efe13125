// NOT OpenCV: see core.hpp in this directory.  Declarations only (opencv_contrib 4.6 aruco).
#ifndef ARVX_TESTS_MOCK_OPENCV_ARUCO_HPP
#define ARVX_TESTS_MOCK_OPENCV_ARUCO_HPP
#include "opencv2/core.hpp"
namespace cv {
namespace aruco {
enum PREDEFINED_DICTIONARY_NAME { DICT_4X4_50 = 0, DICT_6X6_250 = 10 };
class Dictionary {};
Ptr<Dictionary> getPredefinedDictionary(PREDEFINED_DICTIONARY_NAME name);
struct DetectorParameters {
    static Ptr<DetectorParameters> create();
};
class Board {
   public:
    Ptr<Dictionary> dictionary;
};
void detectMarkers(InputArray image, const Ptr<Dictionary> &dictionary, OutputArrayOfArrays corners,
                   OutputArray ids, const Ptr<DetectorParameters> &parameters = DetectorParameters::create(),
                   OutputArrayOfArrays rejectedImgPoints = noArray());
void refineDetectedMarkers(InputArray image, const Ptr<Board> &board, InputOutputArrayOfArrays detectedCorners,
                           InputOutputArray detectedIds, InputOutputArrayOfArrays rejectedCorners,
                           InputArray cameraMatrix = noArray(), InputArray distCoeffs = noArray());
void drawDetectedMarkers(InputOutputArray image, InputArrayOfArrays corners, InputArray ids = noArray(),
                         Scalar borderColor = Scalar(0, 255, 0));
double calibrateCameraAruco(InputArrayOfArrays corners, InputArray ids, InputArray counter,
                            const Ptr<Board> &board, Size imageSize, InputOutputArray cameraMatrix,
                            InputOutputArray distCoeffs, OutputArrayOfArrays rvecs = noArray(),
                            OutputArrayOfArrays tvecs = noArray(), int flags = 0);
}  // namespace aruco
}  // namespace cv
#endif
