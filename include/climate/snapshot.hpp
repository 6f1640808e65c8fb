// include/climate/snapshot.hpp — NetCDF snapshot output with the reference's function names and
// file layout (reference include/io.hpp:70-81, src/io.cpp:378-448) but without PnetCDF: the
// classic CDF-5 (NC_64BIT_DATA) container the reference creates is a simple documented binary
// format — big-endian header {magic, numrecs, dims time(UNLIMITED)/y/x, 7 global text
// attributes, variable u(time,y,x) double} followed by one record per snapshot.  Every rank
// writes its own block with pwrite(); rank 0 writes the header and the record count.
// A reader for the same container serves the IC-from-file extension (the reference
// itself throws for ic.mode=file).
#pragma once
#include <string>
#include <vector>

#include "config.hpp"
#include "core.hpp"

constexpr int NC_NOERR = 0;

int open_netcdf_parallel(const std::string& filename, const Decomp2D& dec, const SimConfig& cfg,
                         MPI_Comm comm, int& ncid, int& varid);
bool write_field_netcdf(int ncid, int varid, const Field& f, const Decomp2D& dec, int step);
// same, from a packed ny_local x nx_local interior (what climate::Stepper::download_interior gives)
bool write_interior_netcdf(int ncid, int varid, const double* interior, const Decomp2D& dec, int step);
void close_netcdf_parallel(int ncid);

// reads record `step` of variable `var` (double, dims [time,]y,x) from a classic CDF-1/2/5 file
void read_netcdf_2d(const std::string& filename, const std::string& var, int step, int& ny, int& nx,
                    std::vector<double>& out);
// a window of that record (rows y0.., columns x0.., wy x wx) read row by row straight into `dst`
// (row stride dst_stride doubles): what each rank of a decomposed run loads of an IC file — no
// buffer of global size anywhere (per-rank start/count like reference src/io.cpp:402-418);
// ny/nx return the variable's full extent
void read_netcdf_window(const std::string& filename, const std::string& var, int step, int y0, int x0,
                        int wy, int wx, double* dst, size_t dst_stride, int& ny, int& nx);
// extent of the variable without reading data
void netcdf_dims_2d(const std::string& filename, const std::string& var, int& ny, int& nx);
// the global text attributes of such a file, in file order
std::vector<std::pair<std::string, std::string>> read_netcdf_attrs(const std::string& filename);
