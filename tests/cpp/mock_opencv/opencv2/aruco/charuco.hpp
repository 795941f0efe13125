// NOT OpenCV: see ../core.hpp in this directory.  Declarations only (opencv_contrib 4.6 aruco).
#ifndef ARVX_TESTS_MOCK_OPENCV_ARUCO_CHARUCO_HPP
#define ARVX_TESTS_MOCK_OPENCV_ARUCO_CHARUCO_HPP
#include "opencv2/aruco.hpp"
namespace cv {
namespace aruco {
class CharucoBoard : public Board {
   public:
    static Ptr<CharucoBoard> create(int squaresX, int squaresY, float squareLength, float markerLength,
                                    const Ptr<Dictionary> &dictionary);
    void draw(Size outSize, OutputArray img, int marginSize = 0, int borderBits = 1);
};
int interpolateCornersCharuco(InputArrayOfArrays markerCorners, InputArray markerIds, InputArray image,
                              const Ptr<CharucoBoard> &board, OutputArray charucoCorners,
                              OutputArray charucoIds, InputArray cameraMatrix = noArray(),
                              InputArray distCoeffs = noArray(), int minMarkers = 2);
void drawDetectedCornersCharuco(InputOutputArray image, InputArray charucoCorners,
                                InputArray charucoIds = noArray(), Scalar cornerColor = Scalar(255, 0, 0));
bool estimatePoseCharucoBoard(InputArray charucoCorners, InputArray charucoIds, const Ptr<CharucoBoard> &board,
                              InputArray cameraMatrix, InputArray distCoeffs, InputOutputArray rvec,
                              InputOutputArray tvec, bool useExtrinsicGuess = false);
double calibrateCameraCharuco(InputArrayOfArrays charucoCorners, InputArrayOfArrays charucoIds,
                              const Ptr<CharucoBoard> &board, Size imageSize, InputOutputArray cameraMatrix,
                              InputOutputArray distCoeffs, OutputArrayOfArrays rvecs, OutputArrayOfArrays tvecs,
                              OutputArray stdDeviationsIntrinsics, OutputArray stdDeviationsExtrinsics,
                              OutputArray perViewErrors, int flags = 0);
}  // namespace aruco
}  // namespace cv
#endif
