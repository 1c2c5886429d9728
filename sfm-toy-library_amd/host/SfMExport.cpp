// SfMExport.cpp -- see SfMExport.h.  Output is byte-for-byte what the reference's std::ofstream insertions produce
// (SfM.cpp:636-710): default floating-point formatting (6 significant digits, %g style), header lines padded with the
// reference's trailing blanks, a blank before every end of line of the vertex records.
#include "SfMExport.h"

#include <cmath>
#include <fstream>

namespace sfmtoylib {

namespace {

// cv::Mat::at<cv::Vec3b>(cv::Point) with a cv::Point2f argument (SfM.cpp:656): the float point converts to an integer point
// through cv::saturate_cast<int> == cvRound (round half to even), x = column, y = row.
void pixelBGR(const ImageBGR& img, const cv::Point2f& p, int bgr[3]) {
    const long col = std::lrint((double)p.x), row = std::lrint((double)p.y);
#ifdef SFMBA_HAVE_OPENCV
    const cv::Vec3b v = img.at<cv::Vec3b>((int)row, (int)col);
    bgr[0] = v[0]; bgr[1] = v[1]; bgr[2] = v[2];
#else
    const size_t o = ((size_t)row * (size_t)img.cols + (size_t)col) * 3;
    bgr[0] = img.data[o]; bgr[1] = img.data[o + 1]; bgr[2] = img.data[o + 2];
#endif
}

}  // namespace

bool SfMExport::saveCloudAndCamerasToPLY(const std::string& prefix, const PointCloud& cloud, const std::vector<cv::Matx34f>& poses,
                                         const std::vector<Features>& features, const std::vector<ImageBGR>& images) {
    std::ofstream pointsFile(prefix + "_points.ply");
    // header, with the reference's column padding (SfM.cpp:639-648)
    pointsFile << "ply                 " << std::endl
               << "format ascii 1.0    " << std::endl
               << "element vertex " << cloud.size() << std::endl
               << "property float x    " << std::endl
               << "property float y    " << std::endl
               << "property float z    " << std::endl
               << "property uchar red  " << std::endl
               << "property uchar green" << std::endl
               << "property uchar blue " << std::endl
               << "end_header          " << std::endl;
    for (const Point3DInMap& point : cloud) {
        // colour: the pixel under the feature of the FIRST originating view (SfM.cpp:652-656)
        const auto firstView = point.originatingViews.begin();
        int bgr[3];
        pixelBGR(images[(size_t)firstView->first], features[(size_t)firstView->first].points[(size_t)firstView->second], bgr);
        pointsFile << point.p.x << " " << point.p.y << " " << point.p.z << " " << bgr[2] << " " << bgr[1] << " " << bgr[0] << " " << std::endl;
    }
    pointsFile.close();

    std::ofstream camerasFile(prefix + "_cameras.ply");
    camerasFile << "ply                 " << std::endl
                << "format ascii 1.0    " << std::endl
                << "element vertex " << (poses.size() * 4) << std::endl
                << "property float x    " << std::endl
                << "property float y    " << std::endl
                << "property float z    " << std::endl
                << "element edge " << (poses.size() * 3) << std::endl
                << "property int vertex1" << std::endl
                << "property int vertex2" << std::endl
                << "property uchar red  " << std::endl
                << "property uchar green" << std::endl
                << "property uchar blue " << std::endl
                << "end_header          " << std::endl;
    // per pose: the translation column and that point moved 0.2 along each column of the rotation part, in double (SfM.cpp:683-693)
    for (const cv::Matx34f& pose : poses) {
        const double c[3] = { (double)pose(0, 3), (double)pose(1, 3), (double)pose(2, 3) };
        camerasFile << c[0] << " " << c[1] << " " << c[2] << std::endl;
        for (int axis = 0; axis < 3; ++axis) {
            const double tip[3] = { c[0] + (double)pose(0, axis) * 0.2, c[1] + (double)pose(1, axis) * 0.2, c[2] + (double)pose(2, axis) * 0.2 };
            camerasFile << tip[0] << " " << tip[1] << " " << tip[2] << std::endl;
        }
    }
    static const char* const axisColour[3] = { "255 0 0", "0 255 0", "0 0 255" };
    for (size_t i = 0; i < poses.size(); i++)
        for (int axis = 0; axis < 3; ++axis) camerasFile << (i * 4) << " " << (i * 4 + 1 + (size_t)axis) << " " << axisColour[axis] << std::endl;
    camerasFile.close();
    return !pointsFile.fail() && !camerasFile.fail();
}

}  // namespace sfmtoylib
