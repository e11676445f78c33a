#include "pointcloud2.hpp"

#include <cstring>
#include <fstream>
#include <sstream>

namespace pcop_host {

bool find_xyz_fields(const PointCloud2& msg, uint32_t* off_x, uint32_t* off_y, uint32_t* off_z, std::string* error) {
    bool fx = false, fy = false, fz = false;
    for (const PointField& f : msg.fields) {
        if (f.datatype != PointField::FLOAT32) continue;
        if (f.name == "x") { *off_x = f.offset; fx = true; }
        else if (f.name == "y") { *off_y = f.offset; fy = true; }
        else if (f.name == "z") { *off_z = f.offset; fz = true; }
    }
    if (!(fx && fy && fz)) {
        if (error) *error = "PointCloud2 has no FLOAT32 x/y/z fields";
        return false;
    }
    // (written so that an offset near 2^32 cannot wrap past the test: the same form as stage_pointcloud2 in csrc/pcop_capi.hip)
    if (msg.point_step < 4 || *off_x > msg.point_step - 4 || *off_y > msg.point_step - 4 || *off_z > msg.point_step - 4) {
        if (error) *error = "x/y/z offsets exceed point_step";
        return false;
    }
    return true;
}

namespace {
float read_f32(const uint8_t* p, bool bigendian) {
    uint32_t w;
    if (bigendian) w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    else w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    float f;
    std::memcpy(&f, &w, 4);
    return f;
}
}  // namespace

bool from_pointcloud2(const PointCloud2& msg, PointCloudXYZ* cloud, std::string* error) {
    uint32_t ox = 0, oy = 0, oz = 0;
    if (!find_xyz_fields(msg, &ox, &oy, &oz, error)) return false;
    const size_t n = (size_t)msg.width * msg.height;
    if (msg.data.size() < (size_t)msg.row_step * msg.height || msg.row_step < (size_t)msg.width * msg.point_step) {
        if (error) *error = "PointCloud2 data shorter than width*height*point_step";
        return false;
    }
    cloud->header = msg.header;
    cloud->width = msg.width; cloud->height = msg.height; cloud->is_dense = msg.is_dense;
    cloud->points.resize(n);
    for (uint32_t r = 0; r < msg.height; ++r) {
        const uint8_t* row = msg.data.data() + (size_t)r * msg.row_step;
        for (uint32_t c = 0; c < msg.width; ++c) {
            const uint8_t* p = row + (size_t)c * msg.point_step;
            PointXYZ& q = cloud->points[(size_t)r * msg.width + c];
            q.x = read_f32(p + ox, msg.is_bigendian); q.y = read_f32(p + oy, msg.is_bigendian); q.z = read_f32(p + oz, msg.is_bigendian);
            q.pad = 1.0f;
        }
    }
    return true;
}

void to_pointcloud2(const PointCloudXYZ& cloud, PointCloud2* msg) {
    msg->header = cloud.header;
    msg->height = 1;
    msg->width = (uint32_t)cloud.points.size();
    msg->fields.clear();
    const char* names[3] = {"x", "y", "z"};
    for (int a = 0; a < 3; ++a) {
        PointField f;
        f.name = names[a]; f.offset = 4u * a; f.datatype = PointField::FLOAT32; f.count = 1;
        msg->fields.push_back(f);
    }
    msg->is_bigendian = false;
    msg->point_step = 16;
    msg->row_step = msg->point_step * msg->width;
    msg->is_dense = cloud.is_dense;
    msg->data.resize((size_t)msg->row_step);
    if (!cloud.points.empty()) std::memcpy(msg->data.data(), cloud.points.data(), msg->data.size());
}

bool write_pointcloud2_file(const std::string& path, const PointCloud2& msg, std::string* error) {
    std::ofstream f(path, std::ios::binary);
    if (!f) { if (error) *error = "cannot write " + path; return false; }
    f << "PCOP_POINTCLOUD2 1\n";
    f << "frame_id " << (msg.header.frame_id.empty() ? "-" : msg.header.frame_id) << "\n";
    f << "height " << msg.height << "\nwidth " << msg.width << "\n";
    f << "is_bigendian " << (msg.is_bigendian ? 1 : 0) << "\npoint_step " << msg.point_step << "\nrow_step " << msg.row_step << "\n";
    f << "is_dense " << (msg.is_dense ? 1 : 0) << "\nfields " << msg.fields.size() << "\n";
    for (const PointField& pf : msg.fields) f << pf.name << " " << pf.offset << " " << (int)pf.datatype << " " << pf.count << "\n";
    f << "data " << msg.data.size() << "\n";
    f.write(reinterpret_cast<const char*>(msg.data.data()), (std::streamsize)msg.data.size());
    return (bool)f;
}

bool read_pointcloud2_file(const std::string& path, PointCloud2* msg, std::string* error) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (error) *error = "cannot open " + path; return false; }
    std::string magic; int ver = 0;
    f >> magic >> ver;
    if (magic != "PCOP_POINTCLOUD2") { if (error) *error = "not a PCOP_POINTCLOUD2 file"; return false; }
    std::string key; size_t nfields = 0, nbytes = 0; int b = 0;
    f >> key >> msg->header.frame_id; if (msg->header.frame_id == "-") msg->header.frame_id.clear();
    f >> key >> msg->height >> key >> msg->width;
    f >> key >> b; msg->is_bigendian = b != 0;
    f >> key >> msg->point_step >> key >> msg->row_step;
    f >> key >> b; msg->is_dense = b != 0;
    f >> key >> nfields;
    // a count no file of this format can hold is a corrupt header, not an allocation to attempt
    if (!f || nfields > 4096) { if (error) *error = "corrupt header in " + path; return false; }
    msg->fields.resize(nfields);
    for (PointField& pf : msg->fields) { int dt = 0; f >> pf.name >> pf.offset >> dt >> pf.count; pf.datatype = (uint8_t)dt; }
    f >> key >> nbytes;
    f.get();   // the newline after the byte count
    if (!f) { if (error) *error = "corrupt header in " + path; return false; }
    {   // the blob cannot be longer than what is left of the file
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end = f.tellg();
        f.seekg(here);
        if (here < 0 || end < here || nbytes > (size_t)(end - here)) { if (error) *error = "truncated file " + path; return false; }
    }
    msg->data.resize(nbytes);
    f.read(reinterpret_cast<char*>(msg->data.data()), (std::streamsize)nbytes);
    if (!f) { if (error) *error = "truncated file " + path; return false; }
    return true;
}

}  // namespace pcop_host
