// hand-written: image_map pigments and uv blocks on both triangle kinds
camera { location <0, 0, 5> up <0, 1, 0> right <1.33333, 0, 0> look_at <0, 0, 0> }
light_source { <3, 4, 5> color rgb <1, 1, 1> }
triangle { <-1, -1, 0>, <1, -1, 0>, <0, 1, 0>
   uv {<0.000000, 0.000000>, <1.000000, 0.000000>, <0.500000, 1.000000>}
   pigment { image_map "checker.bmp" }
   finish { ambient 0.3 diffuse 0.6 }
}
triangle { <-3, -1, -1>, <-1, -1, -1>, <-2, 1, -1>
   uv {<0.25, 0.25>, <0.75, 0.25>, <0.5, 0.75>}
   pigment { image_map "other.bmp" }
}
triangle { <1, -1, -1>, <3, -1, -1>, <2, 1, -1>
   uv {<0, 0>, <2, 0>, <1, 2>}
   pigment { image_map "checker.bmp" }
}
triangle { <-1, 1.2, 0>, <1, 1.2, 0>, <0, 2, 0>
   pigment { color rgb <0.2, 0.9, 0.3> }
}
smooth_triangle { <-1, -3, 0>, <0, 0, 1>, <1, -3, 0>, <0, 0, 1>, <0, -1.2, 0>, <0, 0.2, 1>
   uv {<0, 0>, <1, 0>, <0.5, 1>}
   pigment { image_map "checker.bmp" }
   finish { ambient 0.2 diffuse 0.7 specular 0.3 roughness 0.05 }
}
